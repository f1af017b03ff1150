"""The connected training step of bench.py (RPN -> sampler -> poolers -> targets -> masked loss -> backward): the path
without any host read (proposal counts on the device: DeviceProposals + label_and_sample_proposals_fixed(limits=...))
gives the SAME BITS as the reference's data flow with its host sync after the NMS (exact-size proposal lists), for the
same random keys -- eagerly and replayed as one HIP graph; and every stage agrees with the oracle
(proposal_generator/rpn.py:431-480, roi_heads/roi_heads.py:220-295, modeling/poolers.py:206)."""
import numpy as np
import pytest
import torch

import bench
from oracle import sampling as osp

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def workload():
    w = bench.Workload(DEV, torch.bfloat16, "nhwc")
    w.connected = True
    return w


def _keys(seed, n_img):
    g = torch.Generator().manual_seed(seed)
    rpn = torch.rand(n_img, 268569, generator=g).to(DEV)
    roi = [torch.rand(1000 + bench.N_GT, generator=g).to(DEV) for _ in range(n_img)]
    return rpn, roi


def _run(w, rpn_keys, roi_keys, sync):
    for f in w.feats:
        f.grad = None
    w.mask_logits.grad = None
    out = bench.connected_forward(w, None, rpn_keys, roi_keys, sync=sync)
    torch.autograd.backward([out["box_features"], out["mask_features"], out["loss"]], [w.gbox, w.gmask, None])
    res = {k: out["sample"][k].clone() for k in ("boxes", "classes", "gt_index", "index", "counts")}
    res.update(anchor_labels=out["anchors"][0].clone(), anchor_counts=out["anchors"][2].clone(),
               box_features=out["box_features"].detach().clone(), mask_features=out["mask_features"].detach().clone(),
               loss=out["loss"].detach().clone(), stats=out["stats"].clone(),
               grads=[f.grad.clone() for f in w.feats], logit_grad=w.mask_logits.grad.clone())
    return res, out


def _same(a, b):
    for k in a:
        if isinstance(a[k], list):
            assert all(torch.equal(x, y) for x, y in zip(a[k], b[k])), k
        else:
            assert torch.equal(a[k], b[k]), k


def test_device_counts_path_equals_the_host_sync_path(workload):
    w = workload
    rpn_keys, roi_keys = _keys(3, w.n_img)
    dev_res, out = _run(w, rpn_keys, roi_keys, sync=False)
    props = out["done"]()  # (the sync the step itself never makes)
    del out
    sync_res, out = _run(w, rpn_keys, roi_keys, sync=True)
    del out
    _same(dev_res, sync_res)
    # the device-side counts are the ones the host path reads
    dp_counts = [min(1000, len(p)) for p in props]
    assert dp_counts == [int(c) for c in bench.connected_forward(w, None, rpn_keys, roi_keys)["done"].device.counts()]
    # the sampler against the oracle on the proposals the host path saw
    for i, p in enumerate(props):
        pb = p.proposal_boxes.tensor.cpu().numpy()
        keys = np.concatenate([roi_keys[i][:len(pb)].cpu().numpy(), roi_keys[i][1000:].cpu().numpy()])
        want = osp.label_and_sample_fixed(pb, len(pb), w.gt[i].cpu().numpy(), w.gt_classes[i].cpu().numpy(), keys,
                                          [0.5], [0, 1], bench.ROI_BATCH, bench.ROI_POS_FRACTION, 80)
        for k in ("counts", "index", "classes", "gt_index", "boxes"):
            assert np.array_equal(dev_res[k][i].cpu().numpy(), want[k]), (k, i)
        assert int(dev_res["counts"][i, 1]) == bench.ROI_BATCH and 0 < int(dev_res["counts"][i, 0]) <= 128
    # anchors: 256 sampled labels per image, positives <= 128, the oracle's choice for the same keys
    import oracle

    for i in range(w.n_img):
        _m, lab = oracle.matcher(oracle.pairwise_iou(w.gt[i].cpu().numpy(), w.anchors.cpu().numpy()), [0.3, 0.7],
                                 [0, -1, 1], True)
        want = osp.subsample_anchor_labels(lab, rpn_keys[i].cpu().numpy(), bench.RPN_BATCH, bench.RPN_POS_FRACTION)
        assert np.array_equal(dev_res["anchor_labels"][i].cpu().numpy(), want), i
        assert int(dev_res["anchor_counts"][i].sum()) == bench.RPN_BATCH
    # the masked loss counts the positives among the 128 mask rows only
    n_fg = int(sum(min(int(c), bench.MASK_ROWS) for c in dev_res["counts"][:, 0]))
    assert int(dev_res["stats"][5]) == n_fg and int(dev_res["stats"][4]) == w.n_img * bench.MASK_ROWS - n_fg


def test_one_graph_replay_equals_the_eager_step(workload):
    """ONE captured graph for the whole step (what bench.py times): replays rewrite every output with the eager
    step's bits; with in-graph torch.rand keys the replays differ from each other only through the keys."""
    w = workload
    rpn_keys, roi_keys = _keys(5, w.n_img)
    want, out = _run(w, rpn_keys, roi_keys, sync=False)
    del out  # (no autograd graph of an eager step may be alive during the capture: bench.GraphedStep._capture)
    holder = {}

    def whole():
        for f in w.feats:
            f.grad = None
        w.mask_logits.grad = None
        out = bench.connected_forward(w, None, rpn_keys, roi_keys)
        torch.autograd.backward([out["box_features"], out["mask_features"], out["loss"]], [w.gbox, w.gmask, None])
        holder["t"] = (out["sample"]["boxes"], out["sample"]["classes"], out["sample"]["counts"], out["anchors"][0],
                       out["box_features"].detach(), out["mask_features"].detach(), out["loss"].detach())
        return holder["t"]

    g, outs = bench.GraphedStep._capture(whole)
    for _ in range(3):
        for f in w.feats:
            f.grad.fill_(7.0)
        outs[0].zero_()
        g.replay()
        torch.cuda.synchronize()
        got = dict(zip(("boxes", "classes", "counts", "anchor_labels", "box_features", "mask_features", "loss"), outs))
        for k, v in got.items():
            assert torch.equal(v, want[k]), k
        assert all(torch.equal(f.grad, x) for f, x in zip(w.feats, want["grads"]))
        assert torch.equal(w.mask_logits.grad, want["logit_grad"])
    # the bench's own graphed step (keys drawn inside the graph): runs, finite loss, 512 rows per image
    step = bench.GraphedConnectedStep(w, None)
    l1 = step().clone()
    l2 = step().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(l1) and torch.isfinite(l2)
