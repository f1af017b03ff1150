"""Graph A / graph B variants: which fork / join layouts pay inside a captured HIP graph."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from detectron2_amd.modeling import find_top_rpn_proposals_fused, mask_rcnn_loss_from_targets  # noqa: E402
from detectron2_amd.streams import fork_join  # noqa: E402
from detectron2_amd.structures import crop_and_resize_batch  # noqa: E402

dev = torch.device("cuda:0")
w = bench.Workload(dev, torch.bfloat16, "nhwc")


def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def cap(fn):
    g, out = bench.GraphedStep._capture(fn)
    return g, out


rpn = lambda: find_top_rpn_proposals_fused(w.anchor_levels, w.rpn_logits, w.rpn_deltas, w.image_sizes, 0.7, 2000, 1000,
                                           0.0, True, defer=True)
lab = lambda: [w.anchor_matcher.match_boxes(w.gt[i], w.anchors) for i in range(w.n_img)]
for name, fn in (("A rpn only", rpn), ("A matcher only", lab), ("A serial", lambda: (rpn(), lab())),
                 ("A fork", lambda: fork_join(rpn, lab)), ("A fork, matcher first", lambda: fork_join(lab, rpn)),
                 ("A fork3 m0 m1 rpn", lambda: fork_join(lambda: w.anchor_matcher.match_boxes(w.gt[0], w.anchors),
                                                         lambda: w.anchor_matcher.match_boxes(w.gt[1], w.anchors), rpn)),
                 ("A fork3 m0 rpn m1", lambda: fork_join(lambda: w.anchor_matcher.match_boxes(w.gt[0], w.anchors), rpn,
                                                         lambda: w.anchor_matcher.match_boxes(w.gt[1], w.anchors)))):
    g, _ = cap(fn)
    print(f"{name:28s} {t(g.replay):7.1f} us", flush=True)

box = lambda: w.box_pooler(w.feats, w.box_lists)
mask = lambda: w.mask_pooler(w.feats, w.mask_lists)


def loss():
    tg = crop_and_resize_batch(w.gt_masks, [b.tensor for b in w.mask_lists], 28, w.fg_gt_index, w.crop_status)
    return mask_rcnn_loss_from_targets(w.mask_logits, w.fg_classes, tg)


pm = lambda: [w.proposal_matcher.match_boxes(w.gt[i], w.props_with_gt[i]) for i in range(w.n_img)]


def bwd(yb, ym, ls):
    for f in w.feats:
        f.grad = None
    w.mask_logits.grad = None
    torch.autograd.backward([yb, ym, ls], [w.gbox, w.gmask, None])


def b_serial():
    pm()
    yb, ym = box(), mask()
    ls, _ = loss()
    bwd(yb, ym, ls)
    return ls.detach()


def b_fork4():
    yb, ym, (ls, _), _ = fork_join(box, mask, loss, pm)
    bwd(yb, ym, ls)
    return ls.detach()


def b_fork2():  # poolers on the main stream, targets + loss + labelling beside them
    (yb, ym), (ls, _) = fork_join(lambda: (box(), mask()), lambda: (pm(), loss())[1])
    bwd(yb, ym, ls)
    return ls.detach()


def fwd_only_serial():
    pm(); box(); mask(); loss()


def fwd_only_fork4():
    fork_join(box, mask, loss, pm)


def b_fork2b():  # targets + loss + labelling on the main stream, poolers beside them
    (ls, _), (yb, ym) = fork_join(lambda: (pm(), loss())[1], lambda: (box(), mask()))
    bwd(yb, ym, ls)
    return ls.detach()


def b_fork3():  # box | mask | targets + loss + labelling
    yb, ym, (ls, _) = fork_join(box, mask, lambda: (pm(), loss())[1])
    bwd(yb, ym, ls)
    return ls.detach()


def b_fork2c():  # box + targets/loss | mask + labelling
    (yb, (ls, _)), (ym, _) = fork_join(lambda: (box(), loss()), lambda: (mask(), pm()))
    bwd(yb, ym, ls)
    return ls.detach()


for name, fn in (("B serial", b_serial), ("B fork4", b_fork4), ("B fork2", b_fork2), ("B fork2b", b_fork2b),
                 ("B fork3", b_fork3), ("B fork2c", b_fork2c)):
    with torch.no_grad() if name.startswith("B fwd") else torch.enable_grad():
        g, _ = cap(fn)
    print(f"{name:28s} {t(g.replay):7.1f} us", flush=True)
