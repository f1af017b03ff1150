"""CPU side of the model-level check (tests/test_gpu_reference_models.py): the reference's package imports from its
bytecode tree with the stubs of tests/_reference_model.py, its GeneralizedRCNN / RetinaNet build, and the plain-torch
restatement of torchvision's roi_align that the "reference" backend uses is pinned to the C oracle (itself pinned to the
reference's known answers, tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

import oracle


@pytest.fixture(scope="module")
def rm():
    import _reference_model as m

    m.install()
    return m


@pytest.mark.parametrize("aligned", [True, False])
@pytest.mark.parametrize("sampling_ratio", [0, 2])
def test_torch_roi_align_restatement_equals_the_oracle(rm, aligned, sampling_ratio):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 20, 28)).astype(np.float32)
    rois = np.array([[0, 3.2, 4.1, 60.7, 50.3], [1, 10, 20, 90, 70], [0, -5, -3, 40, 30], [1, 50.5, 10.25, 52.0, 11.0],
                     [0, 0, 0, 111, 79], [1, 100, 60, 130, 95]], np.float32)
    for out, scale in (((7, 7), 0.25), ((14, 14), 0.25), ((3, 5), 0.125)):
        got = rm.torch_roi_align(torch.from_numpy(x), torch.from_numpy(rois), out, scale, sampling_ratio, aligned).numpy()
        exp = oracle.roi_align_forward(x, rois, out, scale, sampling_ratio, aligned)
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() <= 2e-5 * max(np.abs(exp).max(), 1.0), (out, float(np.abs(got - exp).max()))


def test_reference_models_build_from_the_bytecode_tree(rm):
    model = rm.build_model(rm.mask_rcnn_cfg(), seed=0, device="cpu")
    assert type(model).__name__ == "GeneralizedRCNN" and type(model).__module__ == "detectron2.modeling.meta_arch.rcnn"
    assert abs(sum(p.numel() for p in model.parameters()) - 44.3e6) < 0.2e6  # Mask R-CNN R50-FPN
    import detectron2.modeling.poolers as poolers

    assert poolers.__file__.endswith(".pyc")  # the reference's own file, not a restatement
    retina = rm.build_model(rm.retinanet_cfg(), seed=0, device="cpu")
    assert type(retina).__name__ == "RetinaNet"


def test_reference_backend_runs_mask_rcnn_on_the_cpu(rm):
    """The whole reference model on the plain-torch / host-oracle operators (no product code involved): finite losses,
    100 detections with masks -- the second leg of the GPU comparison works on its own."""
    from detectron2.utils.events import EventStorage

    cfg = rm.mask_rcnn_cfg()
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.0
    model = rm.build_model(cfg, seed=0, device="cpu")
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "conv3") and hasattr(m.conv3, "norm"):
                m.conv3.norm.weight.fill_(0.2)
    inputs = rm.make_inputs(1, (192, 256), 4, seed=1, device="cpu")
    with rm.backend("reference"), EventStorage(0):
        model.train()
        torch.manual_seed(0)
        losses = model(inputs)
    assert set(losses) == {"loss_cls", "loss_box_reg", "loss_mask", "loss_rpn_cls", "loss_rpn_loc"}
    assert all(np.isfinite(float(v.detach())) for v in losses.values())
    model.eval()
    with rm.backend("reference"), torch.no_grad():
        out = model(inputs)[0]["instances"]
    assert len(out) == 100 and out.pred_masks.shape == (100, 192, 256)
