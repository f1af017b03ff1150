"""d2amd_label_and_sample_proposals (csrc/label_sample.hip) against oracle/sampling.py: bit-exact (indices, classes,
matched ground truth, counts, boxes) for the same keys; the mask loss over the rows that count
(d2amd_mask_rcnn_loss_forward_masked / _backward_masked) against oracle/mask_head.py.
First run on a GPU at the start of round 3 (profiles/r03/first_run_label_sample_masked_loss.log: 8 passed)."""
import os

import numpy as np
import pytest
import torch

from oracle import sampling as osp

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)
CASES = ["typical", "no_gt", "few", "many_positives", "ignore_band"]


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def _cfg(g, name):
    c = g[f"{name}_cfg"]
    t = int(c[0])
    return [float(v) for v in c[1:1 + t]], [int(v) for v in c[1 + t:2 + 2 * t]], int(c[-2]), float(c[-1])


def _same(got, want, i):
    for k in ("counts", "index", "classes", "gt_index", "boxes"):
        g = got[k][i].cpu().numpy()
        assert np.array_equal(g, want[k]), (k, i)
    # the same rows in pooler format (image index, box): what ROIPooler.pool_rois takes
    S = got["boxes"].shape[1]
    r = got["rois"].view(-1, S, 5)[i].cpu().numpy()
    assert np.array_equal(r[:, 1:], want["boxes"]) and (r[:, 0] == i).all(), i
    if "head_rois" in got:
        H = got["head_rois"].shape[0] // got["boxes"].shape[0]
        h = got["head_rois"].view(-1, H, 5)[i].cpu().numpy()
        assert np.array_equal(h[:, 1:], want["boxes"][:H]) and (h[:, 0] == i).all(), i


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "label_sample.npz"))


def test_batch_of_golden_cases(golden):
    """All golden cases as ONE batch (different sizes, one image without ground truth), each with a device-side limit
    on its proposal count; tied keys; two configurations of the matcher need two calls."""
    from detectron2_amd.modeling import label_and_sample_proposals_fixed

    rng = np.random.default_rng(11)
    for group in (["typical", "no_gt", "few"], ["many_positives"], ["ignore_band"]):
        thr, lab, S, frac = _cfg(golden, group[0])
        props, gts, gcs, keys, lims, ns = [], [], [], [], [], []
        for name in group:
            assert _cfg(golden, name) == (thr, lab, S, frac)
            p, g, gc = golden[f"{name}_proposals"], golden[f"{name}_gt"], golden[f"{name}_gt_classes"]
            k = rng.random(len(p) + len(g), dtype=np.float32)
            if len(p) >= 80:
                k[:40] = k[40:80]
            n = len(p) - (len(p) // 7 if name != "few" else 0)  # fewer valid rows than the buffer holds
            props.append(p); gts.append(g); gcs.append(gc); keys.append(k); ns.append(n)
            lims.append(np.array([n, len(p) + 5], np.int64))
        out = label_and_sample_proposals_fixed([cu(p) for p in props], [cu(g).reshape(-1, 4) for g in gts],
                                               [cu(c) for c in gcs], limits=[cu(l) for l in lims],
                                               keys=[cu(k) for k in keys], thresholds=thr, labels=lab,
                                               batch_size_per_image=S, positive_fraction=frac, num_classes=80,
                                               head_rows=S // 4)
        for i in range(len(group)):
            want = osp.label_and_sample_fixed(props[i], ns[i], gts[i], gcs[i], keys[i], thr, lab, S, frac, 80)
            _same(out, want, i)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_batches(seed):
    """RPN-sized inputs: 1,000-2,000 proposals + up to 600 ground-truth boxes (two LDS chunks), more images than one
    launch holds (16), no limits, no appended ground truth for odd seeds."""
    from detectron2_amd.modeling import label_and_sample_proposals_fixed

    rng = np.random.default_rng(100 + seed)
    n_img = 18 if seed == 0 else 3
    props, gts, gcs, keys = [], [], [], []
    for i in range(n_img):
        G = int(rng.integers(0, 600 if seed == 2 else 40))
        n = int(rng.integers(1, 2000))
        g = rng.uniform(0, 900, (G, 4)).astype(np.float32)
        g[:, 2:] = g[:, :2] + rng.uniform(20, 300, (G, 2)).astype(np.float32)
        p = rng.uniform(0, 1000, (n, 4)).astype(np.float32)
        p[:, 2:] = p[:, :2] + rng.uniform(5, 320, (n, 2)).astype(np.float32)
        if G:
            k = n // 2
            p[:k] = g[rng.integers(0, G, k)] + rng.normal(0, 10, (k, 4)).astype(np.float32)
        props.append(p); gts.append(g); gcs.append(rng.integers(0, 80, G).astype(np.int64))
        keys.append(rng.random(n + G, dtype=np.float32))
    append = seed % 2 == 0
    out = label_and_sample_proposals_fixed([cu(p) for p in props], [cu(g).reshape(-1, 4) for g in gts],
                                           [cu(c) for c in gcs], keys=[cu(k) for k in keys],
                                           proposal_append_gt=append)
    for i in range(n_img):
        want = osp.label_and_sample_fixed(props[i], len(props[i]), gts[i], gcs[i], keys[i], append_gt=append)
        _same(out, want, i)


def test_keys_drawn_inside_the_kernel():
    """keygen=DeviceKeyGenerator: the sampler draws its keys itself from the generator's device-resident state.  Same
    results as with the explicit keys of one `uniform` draw at that state (image i's keys = the outputs behind those
    of the images before it), also for more images than one launch holds and inside a replayed graph; the generator
    advances by one draw per call.  (The generator itself is pinned to Philox4x32-10's known answers in
    tests/test_sampling.py / test_gpu_subsample.py.)"""
    from detectron2_amd.modeling import DeviceKeyGenerator, label_and_sample_proposals_fixed
    from oracle.sampling import philox_uniform_keys

    rng = np.random.default_rng(11)
    for n_img in (2, 19):
        props, gts, gcs = [], [], []
        for i in range(n_img):
            G, n = int(rng.integers(0, 30)), int(rng.integers(1, 1500))
            g = rng.uniform(0, 900, (G, 4)).astype(np.float32)
            g[:, 2:] = g[:, :2] + rng.uniform(20, 300, (G, 2)).astype(np.float32)
            p = rng.uniform(0, 1000, (n, 4)).astype(np.float32)
            p[:, 2:] = p[:, :2] + rng.uniform(5, 320, (n, 2)).astype(np.float32)
            if G:
                p[:n // 2] = g[rng.integers(0, G, n // 2)] + rng.normal(0, 10, (n // 2, 4)).astype(np.float32)
            props.append(cu(p)); gts.append(cu(g).reshape(-1, 4)); gcs.append(cu(rng.integers(0, 80, G).astype(np.int64)))
        sizes = [int(p.shape[0] + g.shape[0]) for p, g in zip(props, gts)]
        gen = DeviceKeyGenerator(DEV, seed=1234 + n_img)
        for draw in range(2):  # two calls: the second one at offset 1
            flat = cu(philox_uniform_keys(1234 + n_img, draw, sum(sizes)))  # the oracle's restatement of that draw
            keys, at = [], 0
            for sz in sizes:
                keys.append(flat[at:at + sz]); at += sz
            want = label_and_sample_proposals_fixed(props, gts, gcs, keys=keys, head_rows=64)
            got = label_and_sample_proposals_fixed(props, gts, gcs, keygen=gen, head_rows=64)
            for k in ("boxes", "classes", "gt_index", "index", "counts", "rois", "head_rois", "head_classes"):
                assert torch.equal(got[k], want[k]), (n_img, draw, k)
            assert gen.state.tolist() == [1234 + n_img, draw + 1, 0]
            # the mask head's rows are the leading rows of every image
            assert torch.equal(got["head_classes"], got["classes"][:, :64])
            assert torch.equal(got["head_rois"].view(n_img, 64, 5), got["rois"].view(n_img, -1, 5)[:, :64])
    # replayed: every replay draws the next keys
    gen = DeviceKeyGenerator(DEV, seed=77)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        label_and_sample_proposals_fixed(props[:2], gts[:2], gcs[:2], keygen=gen)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = label_and_sample_proposals_fixed(props[:2], gts[:2], gcs[:2], keygen=gen)
    seen = []
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(out["index"].clone())
        assert int(gen.state[1]) == 2 + rep
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_agrees_with_the_two_step_path():
    """Same labels as Matcher.match_boxes on the concatenated candidates (the fused matcher the eager step uses)."""
    from detectron2_amd.modeling import Matcher, label_and_sample_proposals_fixed

    rng = np.random.default_rng(3)
    g = rng.uniform(0, 600, (9, 4)).astype(np.float32)
    g[:, 2:] = g[:, :2] + rng.uniform(30, 300, (9, 2)).astype(np.float32)
    p = g[rng.integers(0, 9, 700)] + rng.normal(0, 25, (700, 4)).astype(np.float32)
    gc = rng.integers(0, 80, 9).astype(np.int64)
    keys = rng.random(709, dtype=np.float32)
    out = label_and_sample_proposals_fixed([cu(p)], [cu(g)], [cu(gc)], keys=[cu(keys)])
    cand = torch.cat([cu(p), cu(g)])
    midx, mlab = Matcher([0.5], [0, 1], allow_low_quality_matches=False).match_boxes(cu(g), cand)
    rows = int(out["counts"][0, 1])
    sel = out["index"][0, :rows]
    assert torch.equal(out["gt_index"][0, :rows], midx[sel])
    cls = cu(gc)[midx]
    cls[mlab == 0] = 80
    assert torch.equal(out["classes"][0, :rows], cls[sel])


def test_too_many_candidates_is_reported():
    from detectron2_amd import _C
    from detectron2_amd.modeling import label_and_sample_proposals_fixed

    cap = _C.lib().d2amd_label_and_sample_max_candidates()
    p = torch.zeros((cap, 4), device=DEV)
    g = torch.zeros((1, 4), device=DEV)
    with pytest.raises(RuntimeError, match="candidates"):
        label_and_sample_proposals_fixed([p], [g], [torch.zeros(1, dtype=torch.int64, device=DEV)])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_mask_loss(dtype):
    """mask_rcnn_loss_from_targets(ignore_invalid_rows=True) on a fixed-size list == the validated loss on the
    foreground rows alone (value, statistics, gradient), zero gradient in the ignored rows; with no ignored row it is
    bit-identical to the unmasked entry; with no row at all the loss is 0."""
    from detectron2_amd.modeling import mask_rcnn_loss_from_targets
    from oracle import mask_head as omh

    rng = np.random.default_rng(23)
    B, C, M = 40, 80, 28
    xn = (rng.standard_normal((B, C, M, M)) * 2).astype(np.float32)
    t = rng.random((B, M, M)) < 0.4
    cls = rng.integers(0, C, B)
    cls[15:30] = C
    cls[30:] = -1
    x = cu(xn).to(dtype).requires_grad_(True)
    loss, stats = mask_rcnn_loss_from_targets(x, cu(cls), cu(t), ignore_invalid_rows=True)
    (loss * 0.5).backward()
    xs = x.detach()[:15].clone().requires_grad_(True)
    want, wstats = mask_rcnn_loss_from_targets(xs, cu(cls[:15]), cu(t[:15]))
    (want * 0.5).backward()
    assert torch.equal(loss, want)  # same rows, same fixed-order reduction
    assert stats.tolist() == wstats.tolist()[:4] + [25, 15]
    assert torch.equal(x.grad[:15], xs.grad) and not bool(x.grad[15:].any())
    ol, _os, rows = omh.mask_rcnn_loss_masked(x.detach().float().cpu().numpy(), cls, t)
    assert rows == 15 and abs(float(loss) - ol) <= 1e-5 * abs(ol)
    # every row counts: the masked entry is the unmasked one
    cls2 = rng.integers(0, C, B)
    a = x.detach().clone().requires_grad_(True)
    b = x.detach().clone().requires_grad_(True)
    la, sa = mask_rcnn_loss_from_targets(a, cu(cls2), cu(t), ignore_invalid_rows=True)
    lb, sb = mask_rcnn_loss_from_targets(b, cu(cls2), cu(t))
    la.backward(); lb.backward()
    assert torch.equal(la, lb) and sa.tolist() == sb.tolist() + [B] and torch.equal(a.grad, b.grad)
    # no row counts
    z = x.detach().clone().requires_grad_(True)
    lz, sz = mask_rcnn_loss_from_targets(z, cu(np.full(B, C)), cu(t), ignore_invalid_rows=True)
    lz.backward()
    assert float(lz) == 0.0 and sz.tolist() == [0, 0, 0, 0, B, 0] and not bool(z.grad.any())
