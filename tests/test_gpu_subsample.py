"""d2amd_subsample_labels (csrc/subsample.hip) through detectron2_amd.modeling.sampling against oracle/sampling.py:
bit-exact for the same keys (index lists, counts, rewritten anchor labels); the reference's contract
(detectron2/modeling/sampling.py:9-54, proposal_generator/rpn.py:287-305) and the distribution on the device."""
import numpy as np
import pytest
import torch

from oracle import sampling as osp

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _labels(rng, n, p_pos, p_ign, bg, int8=False):
    """positives: classes 1..79 (int8: 1), never equal to bg (80 or 0); ignored: -1; the rest bg."""
    r = rng.random(n)
    lab = np.where(r < p_pos, rng.integers(1, 80, n) if not int8 else 1, np.where(r < p_pos + p_ign, -1, bg))
    return lab.astype(np.int8 if int8 else np.int64)


def _check_batch(lab, keys, num, frac, bg):
    from detectron2_amd.modeling import subsample_labels_batch

    pos, neg, counts = subsample_labels_batch(cu(lab), num, frac, bg, keys=cu(keys))
    pos, neg, counts = pos.cpu().numpy(), neg.cpu().numpy(), counts.cpu().numpy()
    assert pos.shape == (lab.shape[0], int(num * frac)) and neg.shape == (lab.shape[0], num)
    for i in range(lab.shape[0]):
        ep, en = osp.subsample_labels_keys(lab[i], keys[i], num, frac, bg)
        assert counts[i].tolist() == [len(ep), len(en)], (i, counts[i], len(ep), len(en))
        assert np.array_equal(pos[i, :len(ep)], ep) and (pos[i, len(ep):] == -1).all(), i
        assert np.array_equal(neg[i, :len(en)], en) and (neg[i, len(en):] == -1).all(), i


@pytest.mark.parametrize("n,num,frac,p_pos,p_ign", [
    (268569, 256, 0.5, 0.0004, 0.01),   # the RPN's anchors: a few dozen positives, almost everything negative
    (268569, 256, 0.5, 0.01, 0.3),      # more positives than int(num * frac)
    (1032, 512, 0.25, 0.2, 0.0),        # ROI heads: 1,000 proposals + GT
    (1032, 512, 0.25, 0.02, 0.0),       # few positives: negatives fill
    (700, 512, 0.25, 0.9, 0.05),        # few negatives: fewer rows than num_samples
    (100, 512, 0.25, 0.5, 0.1),         # fewer elements than samples
    (5000, 64, 1.0, 0.3, 0.1), (5000, 64, 0.0, 0.3, 0.1),  # only positives can fill / no positive allowed
    (4096, 8, 0.5, 0.0, 0.0), (4097, 8, 0.5, 1.0, 0.0),     # a group is empty; chunk boundary of the selection
])
def test_index_lists_match_the_oracle(n, num, frac, p_pos, p_ign):
    rng = np.random.default_rng(n + num)
    N = 3 if n > 100000 else 5
    lab = np.stack([_labels(rng, n, p_pos, p_ign, 80) for _ in range(N)])
    keys = rng.random((N, n), dtype=np.float32)
    _check_batch(lab, keys, num, frac, 80)


def test_tied_keys_zero_signs_and_int8_labels():
    """Heavily quantised keys (ties decided by the index), +0 / -0 keys, key 1.0, int8 labels with bg 0."""
    rng = np.random.default_rng(12)
    n = 30000
    lab = np.stack([_labels(rng, n, 0.05, 0.2, 0, int8=True) for _ in range(2)])
    keys = (rng.integers(0, 16, (2, n)) / 16.0).astype(np.float32)
    keys[0, ::7] = -0.0
    keys[1, ::5] = 1.0
    _check_batch(lab, keys, 256, 0.5, 0)
    _check_batch(lab, np.zeros_like(keys), 256, 0.5, 0)  # every key equal: the lowest indices of each group


def test_anchor_labels_rewritten_in_place_like_rpn():
    """RPN._subsample_labels for the batch: the Matcher's int8 labels become -1 / 0 / 1 in place, no host sync."""
    from detectron2_amd.modeling import subsample_anchor_labels_

    rng = np.random.default_rng(3)
    N, n = 2, 268569
    lab = np.stack([_labels(rng, n, 0.0003 * (i + 1), 0.02, 0, int8=True) for i in range(N)])
    keys = rng.random((N, n), dtype=np.float32)
    t = cu(lab)
    out, counts = subsample_anchor_labels_(t, 256, 0.5, keys=cu(keys))
    assert out.data_ptr() == t.data_ptr()
    got = out.cpu().numpy()
    for i in range(N):
        exp = osp.subsample_anchor_labels(lab[i], keys[i], 256, 0.5)
        assert np.array_equal(got[i], exp), i
        assert counts[i].tolist() == [int((exp == 1).sum()), int((exp == 0).sum())]
        assert counts[i].sum().item() == 256


@pytest.mark.parametrize("n_pos,n_neg,num,frac,exp", [
    (300, 5000, 512, 0.25, (128, 384)), (20, 5000, 512, 0.25, (20, 492)), (300, 100, 512, 0.25, (128, 100)),
    (0, 50, 64, 0.5, (0, 50)), (10, 0, 64, 0.5, (10, 0)), (0, 0, 64, 0.5, (0, 0)),
])
def test_reference_signature_contract(n_pos, n_neg, num, frac, exp):
    """subsample_labels(labels, num_samples, positive_fraction, bg_label) -> (pos_idx, neg_idx), torch's RNG."""
    from detectron2_amd.modeling import subsample_labels

    g = torch.Generator().manual_seed(0)
    lab = torch.cat([torch.randint(1, 80, (n_pos,), generator=g), torch.full((n_neg,), 0), torch.full((40,), -1)])
    lab = lab[torch.randperm(lab.numel(), generator=g)].to(DEV)
    pos, neg = subsample_labels(lab, num, frac, 0)
    assert (len(pos), len(neg)) == exp and pos.dtype == neg.dtype == torch.int64 and pos.device == lab.device
    assert len(set(pos.tolist())) == len(pos) and len(set(neg.tolist())) == len(neg)
    assert bool(((lab[pos] != -1) & (lab[pos] != 0)).all()) and bool((lab[neg] == 0).all())
    e = subsample_labels(torch.zeros(0, dtype=torch.int64, device=DEV), 16, 0.5, 0)
    assert len(e[0]) == len(e[1]) == 0


def test_uniform_and_reproducible_on_the_device():
    from detectron2_amd.modeling import subsample_labels, subsample_labels_batch

    g0 = torch.Generator().manual_seed(5)
    lab = torch.cat([torch.randint(0, 80, (40,), generator=g0), torch.full((200,), 80), torch.full((10,), -1)])
    lab = lab[torch.randperm(lab.numel(), generator=g0)]
    runs = 3000
    gen = torch.Generator(device=DEV).manual_seed(7)
    batch = lab.to(DEV)[None].expand(runs, -1).contiguous()  # 3,000 independent draws as one batch
    pos, neg, counts = subsample_labels_batch(batch, 32, 0.25, 80, generator=gen)
    assert bool((counts == torch.tensor([8, 24], device=DEV, dtype=torch.int32)).all())
    hits = torch.zeros(lab.numel(), device=DEV)
    assert bool((pos >= 0).all()) and bool((neg[:, :24] >= 0).all()) and bool((neg[:, 24:] == -1).all())
    for idx in (pos.reshape(-1), neg[:, :24].reshape(-1)):
        hits.index_add_(0, idx, torch.ones(idx.numel(), device=DEV))
    hits = hits.cpu()
    is_pos, is_neg = (lab != -1) & (lab != 80), lab == 80
    assert (hits[is_pos] / runs - 8 / 40).abs().max() < 0.04 and (hits[is_neg] / runs - 24 / 200).abs().max() < 0.03
    assert hits[lab == -1].sum() == 0
    d = lab.to(DEV)
    a = subsample_labels(d, 32, 0.25, 80, generator=torch.Generator(device=DEV).manual_seed(3))
    b = subsample_labels(d, 32, 0.25, 80, generator=torch.Generator(device=DEV).manual_seed(3))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_device_key_generator_matches_the_philox_restatement_and_advances_on_the_device():
    """d2amd_uniform_keys (csrc/random_keys.hip) == oracle.sampling.philox_uniform_keys (pinned to the published Philox
    known answers on the CPU) bit for bit; the offset advances in the kernel: consecutive calls and consecutive REPLAYS of
    a captured graph draw the keys of offsets 0, 1, 2, ... with no host involvement."""
    from detectron2_amd.modeling import DeviceKeyGenerator

    g = DeviceKeyGenerator(DEV, seed=987654321)
    n = 2 * (268569 + 1016)
    a, b = g.uniform(n).cpu().numpy(), g.uniform(3, 5).cpu().numpy()
    assert np.array_equal(a, osp.philox_uniform_keys(987654321, 0, n))
    assert np.array_equal(b.reshape(-1), osp.philox_uniform_keys(987654321, 1, 15)) and b.shape == (3, 5)
    assert a.min() >= 0.0 and a.max() < 1.0 and abs(a.mean() - 0.5) < 2e-3
    assert g.state.tolist() == [987654321, 2, 0]
    # captured: every replay sees the next offset
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        g.uniform(1000)
    cur.wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = g.uniform(1000)
    base = int(g.state[1].item())
    for r in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), osp.philox_uniform_keys(987654321, base + r, 1000)), r
    # seeded from torch's generator: reproducible under torch.manual_seed
    torch.manual_seed(5)
    k1 = DeviceKeyGenerator(DEV).uniform(64)
    torch.manual_seed(5)
    k2 = DeviceKeyGenerator(DEV).uniform(64)
    assert torch.equal(k1, k2)
