"""Fork / join of independent op sequences on HIP streams.

The detection hot path has branches that do not depend on each other inside one iteration -- the RPN's proposal
selection + NMS and its anchor labelling (proposal_generator/rpn.py:431-480 computes the two from the same head
outputs, in either order); the box head's and the mask head's pooling and the mask targets
(roi_heads/roi_heads.py:700-760).  The reference issues them one after the other on one stream.  Several of these
kernels cannot fill 256 CUs on their own (the NMS reduction is 10 workgroups, a sort 10, a compaction 2): run on
separate streams the device overlaps them with their neighbours -- also inside a captured HIP graph, where the fork /
join events become graph edges and cost nothing at replay (scripts/probes/probe_graph_branches.hip: two 100 us
kernels of 8 or 256 workgroups take 132 us as forked graph branches, 208 us on one stream).

    labels, proposals_done = fork_join(lambda: [matcher.match_boxes(gt_i, anchors) for gt_i in gt_boxes],
                                       lambda: find_top_rpn_proposals_fused(..., defer=True))

What pays is putting SMALL-grid chains beside something else; kernels that fill the chip on their own (the two
poolers, the mask-target rasteriser) only slow each other down when forked (bench.py: roi_branches / rpn_branches hold
the measured layouts).  Which branch stays on the current stream matters too (branch 0 does).

Everything a branch allocates is allocated on ITS stream: the results may be used on the current stream after the join
(it waits for every branch), and they must be kept alive by the caller until that work is enqueued (the usual rule of
`Tensor.record_stream`; the results are recorded for the current stream here)."""
from typing import Callable, List

import torch

_POOL = {}


def _streams(device: torch.device, n: int, late: bool = False) -> List["torch.cuda.Stream"]:
    # (branches whose join is deferred get streams of their own: a later fork_join must not queue behind them)
    pool = _POOL.setdefault((device.index, late), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _record(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream)
    elif hasattr(obj, "tensor") and isinstance(obj.tensor, torch.Tensor):
        _record(obj.tensor, stream)


def fork_join(*branches: Callable[[], object], device: torch.device = None, current_first: bool = False,
              defer_join: bool = False) -> list:
    """Run branch 0 on the current stream and every other branch on its own side stream, all starting from the
    current point of the current stream; return their results once everything has been ENQUEUED (no host sync): work
    enqueued on the current stream afterwards sees all of it.

    The side branches are enqueued before branch 0 unless `current_first`.  The order matters to a captured graph too:
    hipGraphLaunch hands the nodes to the device in capture order, a couple of microseconds apiece, so the branch
    captured last starts that much later -- capture the critical branch first.

    defer_join: do not join here; the returned list gets one more element, a callable `join()` that makes the
    CURRENT stream (at the time it is called) wait for the side branches -- for a side branch whose results are only
    needed much later than branch 0's (the anchor labelling beside the NMS: the ROI heads need the proposals, nobody
    needs the anchor labels before the losses).  The side branches' results must not be touched before `join()`."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    cur = torch.cuda.current_stream(device)
    if len(branches) <= 1:
        return [b() for b in branches] + ([lambda: None] if defer_join else [])
    side = _streams(device, len(branches) - 1, late=defer_join)
    for st in side:
        st.wait_stream(cur)  # fork: the branch sees everything enqueued so far
    out = [None] * len(branches)
    if current_first:
        out[0] = branches[0]()
    for i, st in enumerate(side, start=1):
        with torch.cuda.stream(st):
            out[i] = branches[i]()
    if not current_first:
        out[0] = branches[0]()

    def join():
        now = torch.cuda.current_stream(device)
        for i, st in enumerate(side, start=1):
            now.wait_stream(st)
            _record(out[i], now)

    if defer_join:
        return out + [join]
    join()
    return out
