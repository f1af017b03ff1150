"""Generates tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Sources of truth (all under /root/reference, never copied):
  * compiled C++ CPU ops (oracle/_ref, built by oracle/build_ref.py from
    detectron2/layers/csrc/{ROIAlignRotated,box_iou_rotated,nms_rotated}/*_cpu.cpp)
  * python modules loaded by file path: detectron2/layers/mask_ops.py, detectron2/structures/boxes.py,
    detectron2/modeling/matcher.py (with a stub for its one `detectron2.layers.nonzero_tuple` import)
Run:  PYTHONPATH=/root/repo python tests/golden/make_golden.py
The .npz files are committed; /root/reference is not needed (and absent) on the GPU box.
"""
import os

import numpy as np
import torch

from oracle import ref

OUT = os.path.dirname(os.path.abspath(__file__))


def rot_boxes(rng, n, scale=100.0, maxwh=60.0):
    b = np.zeros((n, 5), np.float32)
    b[:, 0] = rng.uniform(0, scale, n)
    b[:, 1] = rng.uniform(0, scale, n)
    b[:, 2] = rng.uniform(1, maxwh, n)
    b[:, 3] = rng.uniform(1, maxwh, n)
    b[:, 4] = rng.uniform(-180, 180, n)
    return b


def main():
    ops = ref.compiled()
    mo, bx = ref.py_mask_ops(), ref.py_boxes()
    rng = np.random.default_rng(1234)

    # --- ROIAlignRotated forward / backward (compiled reference) ---
    N, C, H, W, K = 2, 3, 20, 24, 64
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    r = np.zeros((K, 6), np.float32)
    r[:, 0] = rng.integers(0, N, K)
    r[:, 1] = rng.uniform(-4, W * 2 + 4, K)
    r[:, 2] = rng.uniform(-4, H * 2 + 4, K)
    r[:, 3] = rng.uniform(0, 40, K)
    r[:, 4] = rng.uniform(0, 40, K)
    r[:, 5] = rng.uniform(-180, 180, K)
    r[:8, 5] = [0, 90, 180, 270, -90, 45, 0, 0]
    r[6, 3:5] = 0  # empty box
    g = rng.standard_normal((K, C, 7, 7)).astype(np.float32)
    d = dict(x=x, rois=r, grad=g)
    for sr in (0, 2):
        d[f"out_sr{sr}"] = ops.roi_align_rotated_forward(
            torch.from_numpy(x), torch.from_numpy(r), 0.5, 7, 7, sr).numpy()
        d[f"gin_sr{sr}"] = ops.roi_align_rotated_backward(
            torch.from_numpy(g), torch.from_numpy(r), 0.5, 7, 7, N, C, H, W, sr).numpy()
    np.savez_compressed(os.path.join(OUT, "roi_align_rotated.npz"), **d)

    # --- box_iou_rotated + nms_rotated (compiled reference) ---
    b1, b2 = rot_boxes(rng, 96), rot_boxes(rng, 80)
    b1[:10, 4] = rng.choice([0, 90, -90, 180, 45], 10)
    b1[:10, :4] = np.round(b1[:10, :4])
    b2[:10] = b1[:10]  # identical pairs -> IoU 1 paths
    b2[10, 2] = 0  # zero-area
    iou = ops.box_iou_rotated(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()
    nb = rot_boxes(rng, 300, scale=120.0)
    sc = rng.permutation(300).astype(np.float32) / 300.0  # distinct scores
    d = dict(b1=b1, b2=b2, iou=iou, nms_boxes=nb, nms_scores=sc)
    for thr in (0.2, 0.5, 0.7):
        d[f"keep_{int(thr * 10)}"] = ops.nms_rotated(torch.from_numpy(nb), torch.from_numpy(sc), thr).numpy()
    np.savez_compressed(os.path.join(OUT, "rotated_iou_nms.npz"), **d)

    # --- pairwise_iou / ioa / intersection (reference python) ---
    p1 = rng.uniform(0, 100, (40, 4)).astype(np.float32)
    p1[:, 2:] += p1[:, :2]
    p2 = rng.uniform(0, 100, (500, 4)).astype(np.float32)
    p2[:, 2:] += p2[:, :2]
    p2[0] = p1[0]
    p2[1] = [10, 10, 10, 20]  # empty
    B1, B2 = bx.Boxes(torch.from_numpy(p1)), bx.Boxes(torch.from_numpy(p2))
    np.savez_compressed(
        os.path.join(OUT, "pairwise_iou.npz"), b1=p1, b2=p2,
        iou=bx.pairwise_iou(B1, B2).numpy(), ioa=bx.pairwise_ioa(B1, B2).numpy(),
        intersection=bx.pairwise_intersection(B1, B2).numpy())

    # --- paste_masks_in_image (reference python, CPU path) ---
    torch.manual_seed(1234)
    n, h, w = 12, 120, 160
    masks = torch.rand(n, 28, 28)
    xy = torch.rand(n, 2) * torch.tensor([w * 0.8, h * 0.8])
    wh = torch.rand(n, 2) * torch.tensor([w * 0.5, h * 0.5]) + 2
    boxes = torch.cat([xy - 5, xy + wh], 1)
    out = mo.paste_masks_in_image(masks, boxes, (h, w), 0.5).numpy()
    out_u8 = mo.paste_masks_in_image(masks, boxes, (h, w), -1).numpy()
    np.savez_compressed(os.path.join(OUT, "paste_masks.npz"), masks=masks.numpy(), boxes=boxes.numpy(),
                        shape=np.array([h, w]), out_bits=np.packbits(out), out_u8=out_u8)
    # --- Matcher on pairwise_iou (reference python: modeling/matcher.py + structures/boxes.py) ---
    mt = ref.py_matcher()
    g = rng.uniform(0, 200, (12, 4)).astype(np.float32)
    g[:, 2:] = g[:, :2] + rng.uniform(10, 120, (12, 2)).astype(np.float32)
    a = rng.uniform(0, 260, (3000, 4)).astype(np.float32)
    a[:, 2:] = a[:, :2] + rng.uniform(4, 150, (3000, 2)).astype(np.float32)
    a[:12] = g                      # exact matches (IoU 1)
    a[12:24] = g                    # ties of the row maximum
    g[11] = [900, 900, 950, 950]    # a ground truth no anchor overlaps: row maximum 0
    a[100] = a[101]                 # duplicate predictions
    q = bx.pairwise_iou(bx.Boxes(torch.from_numpy(g)), bx.Boxes(torch.from_numpy(a)))
    d = dict(gt=g, boxes=a, quality=q.numpy())
    cases = {"rpn": ([0.3, 0.7], [0, -1, 1], True), "roi": ([0.5], [0, 1], False),
             "retina": ([0.4, 0.5], [0, -1, 1], True), "three": ([0.2, 0.4, 0.6], [-1, 0, -1, 1], False)}
    for name, (thr, lab, low) in cases.items():
        m, l = mt.Matcher(thr, lab, allow_low_quality_matches=low)(q)
        d[f"{name}_matches"], d[f"{name}_labels"] = m.numpy(), l.numpy()
        d[f"{name}_cfg"] = np.array([len(thr)] + thr + lab + [int(low)], np.float64)
    m0, l0 = mt.Matcher([0.3, 0.7], [0, -1, 1], True)(torch.zeros(0, 7))
    d["empty_matches"], d["empty_labels"] = m0.numpy(), l0.numpy()
    np.savez_compressed(os.path.join(OUT, "matcher.npz"), **d)
    # --- RPN proposals: Box2BoxTransform.apply_deltas + find_top_rpn_proposals (reference python) ---
    import oracle
    br = ref.py_box_regression()

    def nms_stub(boxes, scores, idxs, thr):  # torchvision is not installed: the restatement stands in
        return torch.from_numpy(oracle.batched_nms(boxes.numpy(), scores.numpy(), idxs.numpy(), thr))

    pu = ref.py_proposal_utils(nms_stub)
    N, sizes, topk = 2, (1200, 300, 75), 200
    H, W = 160, 208
    anchors, logits, deltas = [], [], []
    for li, a_l in enumerate(sizes):
        s0 = 16.0 * 2 ** li
        c = rng.uniform(0, [W, H], (a_l, 2))
        wh_ = s0 * np.exp(rng.uniform(-0.4, 0.4, (a_l, 2)))
        anchors.append(np.concatenate([c - wh_ / 2, c + wh_ / 2], 1).astype(np.float32))
        logits.append(rng.standard_normal((N, a_l)).astype(np.float32))
        d = (rng.standard_normal((N, a_l, 4)) * [0.3, 0.3, 0.5, 0.5]).astype(np.float32)
        d[0, :3, 2] = 9.0          # exercises the scale clamp
        d[1, 5:9] = [[-30, 0, 0, 0], [0, 0, -8, -8], [40, 40, 0, 0], [0, 0, 0, 0]]  # clipped away / tiny boxes
        deltas.append(d)
    tr = br.Box2BoxTransform(weights=(1.0, 1.0, 1.0, 1.0))
    props = [torch.stack([tr.apply_deltas(torch.from_numpy(d[i]), torch.from_numpy(a)) for i in range(N)])
             for a, d in zip(anchors, deltas)]
    res = pu.find_top_rpn_proposals(props, [torch.from_numpy(l) for l in logits], [(H, W)] * N, 0.7, topk, 300, 2.0, False)
    d = dict(image_hw=np.array([H, W]), pre_nms_topk=np.array(topk), post_nms_topk=np.array(300),
             nms_thresh=np.array(0.7), min_box_size=np.array(2.0))
    for li in range(len(sizes)):
        d[f"anchors{li}"], d[f"logits{li}"], d[f"deltas{li}"] = anchors[li], logits[li], deltas[li]
        d[f"decoded{li}"] = props[li].numpy()
    for i, r in enumerate(res):
        d[f"boxes_img{i}"] = r.proposal_boxes.tensor.numpy()
        d[f"scores_img{i}"] = r.objectness_logits.numpy()
    np.savez_compressed(os.path.join(OUT, "rpn_proposals.npz"), **d)
    print("golden vectors written to", OUT)


class _FakeBoxes:
    def __init__(self, n):
        self.tensor = torch.zeros(n, 4)

    def __len__(self):
        return len(self.tensor)


class _FakeMasks:
    """gt_masks whose crop_and_resize returns prepared (n, M, M) bool targets (the rasteriser itself is pinned
    separately: BitMasks.crop_and_resize goes through torchvision's roi_align, not installed here)."""

    def __init__(self, targets):
        self.targets = targets

    def __len__(self):
        return len(self.targets)

    def crop_and_resize(self, boxes, side):
        assert side == self.targets.shape[-1]
        return self.targets


def mask_head_golden():
    """mask_rcnn_loss / mask_rcnn_inference of detectron2/modeling/roi_heads/mask_head.py run on CPU (fp32)."""
    mh = ref.py_mask_head()
    rng = np.random.default_rng(77)
    d = {}
    for name, per_img, C, M in (("a", [20, 0, 17], 5, 14), ("b", [6], 20, 28), ("agn", [9, 4], 1, 7)):
        B = sum(per_img)
        logits = (rng.standard_normal((B, C, M, M)) * 3).astype(np.float32)
        logits.reshape(-1)[:6] = [0.0, -0.0, 40.0, -40.0, 90.0, -90.0]  # threshold / saturation cases
        cls = rng.integers(0, C, B).astype(np.int64)
        gt = rng.random((B, M, M)) < 0.4
        inst, pred_inst, o = [], [], 0
        for n in per_img:
            i = mh._d2_Instances((M, M))
            i.gt_classes = torch.from_numpy(cls[o:o + n])
            i.proposal_boxes = _FakeBoxes(n)
            i.gt_masks = _FakeMasks(torch.from_numpy(gt[o:o + n]))
            inst.append(i)
            q = mh._d2_Instances((M, M))
            q.pred_classes = torch.from_numpy(cls[o:o + n])
            pred_inst.append(q)
            o += n
        x = torch.from_numpy(logits).requires_grad_(True)
        loss = mh.mask_rcnn_loss(x, inst)
        (loss * 1.75).backward()
        ev = mh._d2_events.scalars
        mh.mask_rcnn_inference(torch.from_numpy(logits), pred_inst)
        probs = torch.cat([q.pred_masks for q in pred_inst]).numpy()
        d.update({f"{name}_logits": logits, f"{name}_classes": cls, f"{name}_gt": gt, f"{name}_per_img": np.array(per_img),
                  f"{name}_loss": loss.detach().numpy(), f"{name}_grad_x1p75": x.grad.numpy(), f"{name}_probs": probs,
                  f"{name}_accuracy": np.array(ev["mask_rcnn/accuracy"]),
                  f"{name}_false_positive": np.array(ev["mask_rcnn/false_positive"]),
                  f"{name}_false_negative": np.array(ev["mask_rcnn/false_negative"])})
    np.savez_compressed(os.path.join(OUT, "mask_head.npz"), **d)
    print("mask_head.npz written")


def dense_detector_golden():
    """DenseDetector._decode_multi_level_predictions of detectron2/modeling/meta_arch/dense_detector.py on CPU, with
    scores = sigmoid(logits) as meta_arch/retinanet.py:267 prepares them."""
    import types

    mod, B2B, Boxes, _ = ref.py_dense_detector()
    rng = np.random.default_rng(2024)
    N, K, sizes = 2, 5, [300, 80, 12]
    thr, topk = 0.3, 50
    dd = mod.DenseDetector.__new__(mod.DenseDetector)
    dd.__dict__["box2box_transform"] = B2B(weights=(1.0, 1.0, 2.0, 2.0))
    anchors, logits, deltas = [], [], []
    for li, a_l in enumerate(sizes):
        s0 = 32.0 * 2 ** li
        c = rng.uniform(0, [320, 256], (a_l, 2))
        wh_ = s0 * np.exp(rng.uniform(-0.4, 0.4, (a_l, 2)))
        anchors.append(np.concatenate([c - wh_ / 2, c + wh_ / 2], 1).astype(np.float32))
        logits.append((rng.standard_normal((N, a_l, K)) * 1.5 - 1.5).astype(np.float32))
        d = (rng.standard_normal((N, a_l, 4)) * [0.3, 0.3, 0.5, 0.5]).astype(np.float32)
        d[0, :2, 2] = 30.0  # exercises the scale clamp
        deltas.append(d)
    d = dict(score_thresh=np.array(thr), topk=np.array(topk), weights=np.array([1.0, 1.0, 2.0, 2.0]))
    for li in range(len(sizes)):
        d[f"anchors{li}"], d[f"logits{li}"], d[f"deltas{li}"] = anchors[li], logits[li], deltas[li]
    for i in range(N):
        scores = [torch.from_numpy(l[i]).clone().sigmoid_() for l in logits]
        for sc in scores:  # the fixture must not depend on topk's order of equal scores
            v = sc[sc > thr]
            assert len(torch.unique(v)) == len(v)
        inst = dd._decode_multi_level_predictions([Boxes(torch.from_numpy(a)) for a in anchors], scores,
                                                  [torch.from_numpy(x[i]) for x in deltas], thr, topk, (256, 320))
        d[f"boxes_img{i}"] = inst.pred_boxes.tensor.numpy()
        d[f"scores_img{i}"] = inst.scores.numpy()
        d[f"classes_img{i}"] = inst.pred_classes.numpy()
    # second case ("t_" keys): HEAVY TIES.  Quantised logits (many equal values), a block of large logits whose fp32
    # sigmoid saturates to the same score although the logits differ, +0 / -0, and a level where the k-th score is
    # tied.  torch.topk's order inside a group of equal scores is unspecified: the consumer
    # (tests/test_oracle_golden.py) requires the same selection outside the tied k-th group and an order that differs
    # only inside groups of equal fp32 score.
    rng = np.random.default_rng(77)
    sizes_t, K_t, thr_t, topk_t = [400, 60, 9], 4, 0.2, 120
    d["t_score_thresh"], d["t_topk"] = np.array(thr_t), np.array(topk_t)
    for li, a_l in enumerate(sizes_t):
        c = rng.uniform(0, [320, 256], (a_l, 2))
        wh_ = 32.0 * 2 ** li * np.exp(rng.uniform(-0.4, 0.4, (a_l, 2)))
        d[f"t_anchors{li}"] = np.concatenate([c - wh_ / 2, c + wh_ / 2], 1).astype(np.float32)
        lg = (np.round(rng.standard_normal((N, a_l, K_t)) * 4) / 2).astype(np.float32)  # steps of 0.5
        lg[:, : a_l // 8, 0] = (17.5 + rng.uniform(0, 8, (N, a_l // 8))).astype(np.float32)  # sigmoid -> 1.0 or 1 - 2^-24
        lg[:, a_l // 8, 1], lg[:, a_l // 8 + 1, 1] = 0.0, -0.0
        d[f"t_logits{li}"] = lg
        d[f"t_deltas{li}"] = (rng.standard_normal((N, a_l, 4)) * [0.3, 0.3, 0.5, 0.5]).astype(np.float32)
    for i in range(N):
        scores = [torch.from_numpy(d[f"t_logits{l}"][i]).clone().sigmoid_() for l in range(3)]
        inst = dd._decode_multi_level_predictions(
            [Boxes(torch.from_numpy(d[f"t_anchors{l}"])) for l in range(3)], scores,
            [torch.from_numpy(d[f"t_deltas{l}"][i]) for l in range(3)], thr_t, topk_t, (256, 320))
        d[f"t_boxes_img{i}"] = inst.pred_boxes.tensor.numpy()
        d[f"t_scores_img{i}"] = inst.scores.numpy()
        d[f"t_classes_img{i}"] = inst.pred_classes.numpy()
        d[f"t_counts_img{i}"] = np.array([min(topk_t, int((sc > thr_t).sum())) for sc in scores])
    np.savez_compressed(os.path.join(OUT, "dense_detector.npz"), **d)
    print("dense_detector.npz written", [len(d[f"scores_img{i}"]) for i in range(N)],
          [len(d[f"t_scores_img{i}"]) for i in range(N)])


def paste_device_path_golden():
    """paste_masks_in_image as the reference evaluates it for DEVICE tensors: `_do_paste_mask(..., skip_empty=False)`
    (mask_ops.py:116-119,134) samples the whole image, so values up to half a mask pixel outside the box are non-zero
    -- visible below threshold 0.5 and in the uint8 soft output.  Run on CPU tensors (same ATen grid_sampler)."""
    mo = ref.py_mask_ops()
    torch.manual_seed(77)
    n, h, w = 6, 150, 200
    masks = torch.rand(n, 28, 28)
    boxes = torch.tensor([[20.3, 10.7, 180.2, 140.9], [-30.0, -20.0, 120.0, 90.0], [60.0, 50.0, 64.5, 53.2],
                          [5.0, 100.0, 195.0, 149.0], [100.0, 0.0, 260.0, 150.0], [150.5, 20.5, 190.5, 140.5]])
    soft, _ = mo._do_paste_mask(masks[:, None], boxes, h, w, skip_empty=False)
    np.savez_compressed(os.path.join(OUT, "paste_masks_full.npz"), masks=masks.numpy(), boxes=boxes.numpy(),
                        shape=np.array([h, w]), out_u8=(soft * 255).to(torch.uint8).numpy(),
                        out_thr01=np.packbits((soft >= 0.1).numpy()), out_thr05=np.packbits((soft >= 0.5).numpy()))
    print("paste_masks_full.npz written", int((soft > 0).sum()))


def label_sample_golden():
    """ROIHeads.label_and_sample_proposals (roi_heads.py:219-295) with the reference's own pairwise_iou, Matcher and
    subsample_labels.  The ROIHeads class itself needs the whole package; the lines around those calls are restated
    here: the concatenation of add_ground_truth_to_proposals (proposal_utils.py:196-203) and the relabelling of
    _sample_proposals (roi_heads.py:199-208).  The random draw of subsample_labels is kept as (sizes, index sets)."""
    import torch

    from oracle import ref

    bx, mt, sp = ref.py_boxes(), ref.py_matcher(), ref.py_sampling()
    rng = np.random.default_rng(20260923)
    torch.manual_seed(7)
    cases = {  # name: (proposals, ground truth, thresholds, labels, batch per image, positive fraction)
        "typical": (1000, 7, [0.5], [0, 1], 512, 0.25),
        "no_gt": (300, 0, [0.5], [0, 1], 512, 0.25),
        "few": (40, 3, [0.5], [0, 1], 512, 0.25),
        "many_positives": (900, 12, [0.3], [0, 1], 256, 0.25),
        "ignore_band": (800, 9, [0.3, 0.7], [0, -1, 1], 512, 0.5),
    }
    d = {}
    for name, (n, G, thr, lab, S, frac) in cases.items():
        g = rng.uniform(0, 600, (G, 4)).astype(np.float32)
        g[:, 2:] = g[:, :2] + rng.uniform(30, 300, (G, 2)).astype(np.float32)
        p = rng.uniform(0, 700, (n, 4)).astype(np.float32)
        p[:, 2:] = p[:, :2] + rng.uniform(8, 320, (n, 2)).astype(np.float32)
        if G:  # jittered copies of the ground truth: the positives
            k = n // 3 if name != "many_positives" else (2 * n) // 3
            src = g[rng.integers(0, G, k)]
            p[:k] = src + rng.normal(0, 12, (k, 4)).astype(np.float32)
            p[k] = g[0]  # an exact duplicate of a ground-truth box
        gc = rng.integers(0, 80, G).astype(np.int64)
        cand = torch.cat([torch.from_numpy(p), torch.from_numpy(g)])  # proposal_utils.py:196-203
        q = bx.pairwise_iou(bx.Boxes(torch.from_numpy(g)), bx.Boxes(cand))
        midx, mlab = mt.Matcher(thr, lab, allow_low_quality_matches=False)(q)
        if G:  # roi_heads.py:199-205
            cls = torch.from_numpy(gc)[midx]
            cls[mlab == 0] = 80
            cls[mlab == -1] = -1
        else:  # :207
            cls = torch.zeros_like(midx) + 80
        pos, neg = sp.subsample_labels(cls, S, frac, 80)
        d.update({f"{name}_proposals": p, f"{name}_gt": g, f"{name}_gt_classes": gc,
                  f"{name}_cfg": np.array([len(thr)] + thr + lab + [S, frac], np.float64),
                  f"{name}_matched_idxs": midx.numpy(), f"{name}_matched_labels": mlab.numpy().astype(np.int8),
                  f"{name}_classes": cls.numpy(), f"{name}_ref_pos": pos.numpy(), f"{name}_ref_neg": neg.numpy()})
        print(name, "candidates", len(cand), "sampled", len(pos), "+", len(neg))
    np.savez_compressed(os.path.join(OUT, "label_sample.npz"), **d)


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "mask_head":
        mask_head_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "paste_full":
        paste_device_path_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "dense_detector":
        dense_detector_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "label_sample":
        label_sample_golden()
    else:
        main()
        mask_head_golden()
        dense_detector_golden()
        label_sample_golden()
