"""detectron2_amd: MI355X (gfx950) implementation of Detectron2's per-image detection hot path.

`detectron2_amd.layers` mirrors the `detectron2.layers` operator surface for that path (same
names, argument meaning, return dtypes, error behaviour); `detectron2_amd.structures` holds
`pairwise_iou` & co.  Everything is backed by hand-written HIP kernels behind the C ABI in
include/d2amd.h (libd2amd.so).  No CPU fallback exists: ops raise on CPU tensors.
"""
__version__ = "0.1"
