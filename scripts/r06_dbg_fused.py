import sys, os, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_reference_models as T
rm = T._rm()
import detectron2
from detectron2_amd import integrate
cfg = rm.mask_rcnn_cfg(); cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.0
model = rm.build_model(cfg, seed=0, device="cuda"); T._tame(model)
inputs = rm.make_inputs(2, (800, 1333), 8, seed=3, device="cuda")
ir = T._infer_pass(rm, model, inputs, "reference", autocast=torch.bfloat16)
import contextlib
for only in (["matcher"], ["pooler"], ["rpn"], ["box_inference"], ["mask_head"], None):
    model.eval()
    with rm.backend("product"), integrate.patch(detectron2, models=[model], layers=False, only=only), torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        out = [o["instances"] for o in model(inputs)]
    print(only, [T._match_detections(a, b) for a, b in zip(out, ir)], flush=True)
