"""HIP-graph capture of the hot path (torch.cuda.CUDAGraph = hipStreamBeginCapture / hipGraphLaunch): every kernel of
libd2amd.so is launched on the caller's stream and the per-call state of the multi-kernel ops is re-zeroed by KERNELS
(graph memset nodes did not replay reliably on ROCm 7.2), so a captured step replays bit-identically -- what
bench.py's default launch mode relies on."""
import numpy as np
import pytest
import torch

from detectron2_amd.layers import batched_nms_images
from detectron2_amd.modeling import Matcher, ROIPooler, find_top_rpn_proposals_fused, mask_rcnn_loss_from_targets
from detectron2_amd.structures import BitMasks, Boxes, crop_and_resize_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _capture(fn):
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        fn()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def test_rpn_half_replays_identically():
    torch.manual_seed(0)
    sizes = [6000, 1500, 400]
    anchors, logits, deltas = [], [], []
    for l, a in enumerate(sizes):
        c = torch.rand(a, 2) * torch.tensor([320.0, 256.0])
        wh = 24.0 * 2 ** l * torch.exp(torch.rand(a, 2) - 0.5)
        anchors.append(torch.cat([c - wh / 2, c + wh / 2], 1).to(DEV))
        logits.append((torch.randn(2, a) + torch.arange(a) * 1e-6).to(DEV))
        deltas.append((torch.randn(2, a, 4) * 0.2).to(DEV))
    hw = [(256, 320)] * 2
    gt = torch.tensor([[10.0, 20, 100, 120], [150, 60, 300, 200]], device=DEV)
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    allanc = torch.cat(anchors)

    def fn():
        done = find_top_rpn_proposals_fused(anchors, logits, deltas, hw, 0.7, 500, 200, 0.0, True, defer=True)
        return done, mt.match_boxes(gt, allanc)

    want = find_top_rpn_proposals_fused(anchors, logits, deltas, hw, 0.7, 500, 200, 0.0, True)
    want_m = mt.match_boxes(gt, allanc)
    g, (done, match) = _capture(fn)
    for _ in range(4):
        g.replay()
        got = done()
        for a, b in zip(got, want):
            assert torch.equal(a.proposal_boxes.tensor, b.proposal_boxes.tensor)
            assert torch.equal(a.objectness_logits, b.objectness_logits)
        assert torch.equal(match[0], want_m[0]) and torch.equal(match[1], want_m[1])


def test_roi_head_half_with_backward_replays_identically():
    rng = np.random.default_rng(4)
    n_img, C = 2, 64
    feats = [torch.from_numpy(rng.standard_normal((n_img, C, 128 // s, 160 // s)).astype(np.float32)).to(DEV)
             .to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in (4, 8, 16, 32)]

    def boxes(k):
        xy = rng.uniform(0, [100, 80], (k, 2))
        wh = rng.uniform(8, 60, (k, 2))
        return torch.from_numpy(np.concatenate([xy, xy + wh], 1).astype(np.float32)).to(DEV)

    bl, ml = [Boxes(boxes(40)) for _ in range(n_img)], [Boxes(boxes(12)) for _ in range(n_img)]
    gb = torch.randn(80, C, 7, 7, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gm = torch.randn(24, C, 14, 14, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    masks = [BitMasks(torch.from_numpy(rng.random((5, 128, 160)) < 0.5).to(DEV)) for _ in range(n_img)]
    idx = [torch.from_numpy(rng.integers(0, 5, 12)).to(DEV) for _ in range(n_img)]
    logit = torch.randn(24, 6, 28, 28, device=DEV).to(torch.bfloat16).requires_grad_(True)
    cls = torch.from_numpy(rng.integers(0, 6, 24)).to(DEV)
    sc = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    p7, p14 = ROIPooler(7, sc, 0, "ROIAlignV2"), ROIPooler(14, sc, 0, "ROIAlignV2")

    def fn():
        yb, ym = p7(feats, bl), p14(feats, ml)
        tg = crop_and_resize_batch(masks, [b.tensor for b in ml], 28, idx)
        loss, _ = mask_rcnn_loss_from_targets(logit, cls, tg)
        for f in feats:
            f.grad = None
        logit.grad = None
        torch.autograd.backward([yb, ym, loss], [gb, gm, None])
        return yb, ym, loss

    yb, ym, loss = fn()
    want = [t.detach().clone() for t in (yb, ym, loss)] + [f.grad.clone() for f in feats] + [logit.grad.clone()]
    # no autograd graph of the eager pass may stay alive: its AccumulateGrad nodes are bound to the default stream and
    # would drag the capture onto it (PyTorch warns; the HIP runtime then crashes at capture end)
    del yb, ym, loss
    g, out = _capture(fn)
    for _ in range(3):
        for f in feats:  # scribble over the outputs: the replay must rewrite all of them
            f.grad.fill_(7.0)
        out[0].zero_()
        g.replay()
        got = [t.detach() for t in out] + [f.grad for f in feats] + [logit.grad]
        for a, b in zip(got, want):
            assert torch.equal(a, b)


def test_large_nms_side_streams_outside_capture_still_work():
    rng = np.random.default_rng(2)
    b = rng.uniform(0, 500, (20000, 4)).astype(np.float32)
    b[:, 2:] = b[:, :2] + rng.uniform(5, 80, (20000, 2)).astype(np.float32)
    s = ((rng.permutation(20000) + 1) / 20001).astype(np.float32)
    c = rng.integers(0, 20, 20000)
    inp = [(torch.from_numpy(b).to(DEV), torch.from_numpy(s).to(DEV), torch.from_numpy(c).to(DEV))] * 2
    k1 = batched_nms_images(inp, 0.5)
    k2 = batched_nms_images(inp, 0.5)
    assert torch.equal(k1[0], k2[0]) and torch.equal(k1[0], k1[1])


def test_forked_branches_eager_and_captured():
    """detectron2_amd.streams.fork_join: the RPN's proposal path and its anchor labelling on two streams -- same
    results as one after the other, eagerly and as the forked branches of a captured graph (bench.py's graph A)."""
    from detectron2_amd.streams import fork_join

    torch.manual_seed(1)
    sizes = [5000, 1200, 300]
    anchors, logits, deltas = [], [], []
    for l, a in enumerate(sizes):
        c = torch.rand(a, 2) * torch.tensor([320.0, 256.0])
        wh = 24.0 * 2 ** l * torch.exp(torch.rand(a, 2) - 0.5)
        anchors.append(torch.cat([c - wh / 2, c + wh / 2], 1).to(DEV))
        logits.append((torch.randn(2, a) + torch.arange(a) * 1e-6).to(DEV))
        deltas.append((torch.randn(2, a, 4) * 0.2).to(DEV))
    hw = [(256, 320)] * 2
    gt = [torch.tensor([[10.0, 20, 100, 120], [150, 60, 300, 200]], device=DEV),
          torch.tensor([[40.0, 40, 90, 200]], device=DEV)]
    mt = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)
    allanc = torch.cat(anchors)
    rpn = lambda: find_top_rpn_proposals_fused(anchors, logits, deltas, hw, 0.7, 400, 150, 0.0, True, defer=True)
    lab = lambda: [mt.match_boxes(g, allanc) for g in gt]
    want_p, want_l = rpn()(), lab()

    def same(props, labels):
        for a, b in zip(props, want_p):
            assert torch.equal(a.proposal_boxes.tensor, b.proposal_boxes.tensor)
            assert torch.equal(a.objectness_logits, b.objectness_logits)
        for (m, l), (wm, wl) in zip(labels, want_l):
            assert torch.equal(m, wm) and torch.equal(l, wl)

    labels, done = fork_join(lab, rpn)
    same(done(), labels)
    g, (labels, done) = _capture(lambda: fork_join(lab, rpn))
    for _ in range(3):
        g.replay()
        same(done(), labels)
    # the same work forked after the selection, beside the NMS only (what bench.py's step does), and branch 0 first
    beside = lambda: find_top_rpn_proposals_fused(anchors, logits, deltas, hw, 0.7, 400, 150, 0.0, True, defer=True,
                                                  beside_nms=lab)
    done = beside()
    same(done(), done.beside)
    g, done = _capture(beside)
    for _ in range(3):
        g.replay()
        same(done(), done.beside)
    done, labels = fork_join(rpn, lab, current_first=True)
    same(done(), labels)
