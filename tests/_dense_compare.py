"""Comparison of a dense-detector selection with the REFERENCE's (torch.topk on fp32 sigmoid scores), shared by the
CPU oracle pin (test_oracle_golden.py) and the GPU golden test (test_gpu_dense.py).

torch.topk leaves the order inside a group of equal scores unspecified, and which members of a tied k-th group it
takes; our side ranks the logit (oracle/dense_detector.py: RANKING RULE).  What must hold, per (image, level):
  * same count, scores equal position by position (to the rounding of exp());
  * for every group of equal reference score: the same SET of (class, box) rows -- i.e. the order differs from the
    reference's only inside such a group;
  * except the last group of a level truncated by topk (a tied k-th score), where only the scores are comparable.
Returns (rows compared by identity, rows in tied k-th groups, rows whose position differs from the reference's)."""
import numpy as np


def assert_same_selection_up_to_ties(got, ref, counts, topk, rtol=2e-6, box_atol=1e-3):
    gb, gs, gc = got
    rb, rs, rc = ref
    assert len(gs) == len(rs) == int(np.sum(counts)), (len(gs), len(rs), counts)
    n_id = n_tail = n_moved = 0
    o = 0
    for n in [int(c) for c in counts]:
        s_ref, s_got = rs[o:o + n], gs[o:o + n]
        assert np.all(np.diff(s_ref) <= 0), "reference scores must be non-increasing inside a level"
        np.testing.assert_allclose(s_got, s_ref, rtol=rtol, atol=0)
        starts = np.flatnonzero(np.r_[True, s_ref[1:] != s_ref[:-1]]) if n else np.zeros(0, np.int64)
        ends = np.r_[starts[1:], n]
        for a, b in zip(starts, ends):
            if b == n and n == topk:  # possibly a tied k-th group: membership is torch.topk's free choice
                n_tail += b - a
                continue
            rows_g = np.c_[gc[o + a:o + b], gb[o + a:o + b]].astype(np.float64)
            rows_r = np.c_[rc[o + a:o + b], rb[o + a:o + b]].astype(np.float64)
            n_moved += int(np.any(np.abs(rows_g - rows_r) > box_atol + rtol * np.abs(rows_r), axis=1).sum())
            kg = np.lexsort(np.round(rows_g, 1).T[::-1])
            kr = np.lexsort(np.round(rows_r, 1).T[::-1])
            assert np.array_equal(rows_g[kg][:, 0], rows_r[kr][:, 0]), "classes of an equal-score group differ"
            np.testing.assert_allclose(rows_g[kg][:, 1:], rows_r[kr][:, 1:], rtol=rtol, atol=box_atol)
            n_id += b - a
        o += n
    return n_id, n_tail, n_moved
