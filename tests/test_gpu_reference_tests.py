"""The REFERENCE's own unit tests of the hot-path operators, executed UNMODIFIED against this package's surface.

/root/reference/tests/{layers/test_roi_align, layers/test_roi_align_rotated, layers/test_nms, layers/test_nms_rotated,
layers/test_deformable, structures/test_rotated_boxes, structures/test_boxes, modeling/test_roi_pooler,
modeling/test_matcher}.py are loaded from the reference tree (or, on the GPU box, from the bytecode oracle/build_ref.py
compiled from those files into oracle/_ref/py/) with `detectron2.*` resolved by tests/_reference_surface.py: the
known answers they hold reach the HIP kernels without a transcription step.  The product has no CPU path: the tests'
CPU tensors are moved to the GPU by the surface's `on_device` wrapper, so their "*_cpu" cases run the same HIP kernels
as their "*_cuda" cases.

Every reference test id is listed below: RUN (executed on the GPU, must pass), RUN_ON_CPU (scripting only, runs where
the reference's source exists) or NOT_RUN (with the reason).  A test asserts that
the lists cover exactly what the reference's files define, so nothing can drop out silently.  The executed / skipped
ids are printed at the end of the session (and written to $D2AMD_REFERENCE_TEST_REPORT when set)."""
import io
import os
import unittest

import pytest
import torch

from conftest import need_reference
from oracle import build_ref, ref

FILES = {
    "roi_align": "tests/layers/test_roi_align.py",
    "roi_align_rotated": "tests/layers/test_roi_align_rotated.py",
    "nms": "tests/layers/test_nms.py",
    "nms_rotated": "tests/layers/test_nms_rotated.py",
    "deformable": "tests/layers/test_deformable.py",
    "rotated_boxes": "tests/structures/test_rotated_boxes.py",
    "boxes": "tests/structures/test_boxes.py",
    "roi_pooler": "tests/modeling/test_roi_pooler.py",
    "matcher": "tests/modeling/test_matcher.py",
}

RUN = [
    "roi_align::ROIAlignTest.test_forward_output",
    "roi_align::ROIAlignTest.test_resize",
    "roi_align::ROIAlignTest.test_grid_sample_equivalence",
    "roi_align::ROIAlignTest.test_empty_box",
    "roi_align::ROIAlignTest.test_empty_batch",
    "roi_align_rotated::ROIAlignRotatedTest.test_forward_output_0_90_180_270",
    "roi_align_rotated::ROIAlignRotatedTest.test_resize",
    "roi_align_rotated::ROIAlignRotatedTest.test_empty_box",
    "roi_align_rotated::ROIAlignRotatedTest.test_roi_align_rotated_gradcheck_cpu",
    "roi_align_rotated::ROIAlignRotatedTest.test_roi_align_rotated_gradient_cuda",
    "nms::TestNMS.test_nms_scriptability",
    "nms_rotated::TestNMSRotated.test_batched_nms_rotated_0_degree_cpu",
    "nms_rotated::TestNMSRotated.test_batched_nms_rotated_0_degree_cuda",
    "nms_rotated::TestNMSRotated.test_nms_rotated_0_degree_cpu",
    "nms_rotated::TestNMSRotated.test_nms_rotated_0_degree_cuda",
    "nms_rotated::TestNMSRotated.test_nms_rotated_90_degrees_cpu",
    "nms_rotated::TestNMSRotated.test_nms_rotated_180_degrees_cpu",
    "deformable::DeformableTest.test_forward_output",
    "deformable::DeformableTest.test_forward_output_on_cpu",
    "deformable::DeformableTest.test_forward_output_on_cpu_equals_output_on_gpu",
    "deformable::DeformableTest.test_small_input",
    "deformable::DeformableTest.test_raise_exception",
    "deformable::DeformableTest.test_repr",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_0_dim_cpu",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_0_dim_cuda",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_half_overlap_cpu",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_half_overlap_cuda",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_precision",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_too_many_boxes_cuda",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_extreme",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_issue_2154",
    "rotated_boxes::TestRotatedBoxesLayer.test_iou_issue_2167",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_0_degree",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_45_degrees",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_orthogonal",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_large_close_boxes",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_many_boxes",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_issue1207_simplified",
    "rotated_boxes::TestRotatedBoxesStructure.test_pairwise_iou_issue1207",
    "boxes::TestBoxIOU.test_pairwise_iou",
    "boxes::TestBoxIOU.test_pairwise_ioa",
    "roi_pooler::TestROIPooler.test_roialignv2_roialignrotated_match_cpu",
    "roi_pooler::TestROIPooler.test_roialignv2_roialignrotated_match_cuda",
    "roi_pooler::TestROIPooler.test_no_images",
]

# TorchScript compiles from SOURCE TEXT: a test that scripts a function / module defined inside the reference's test file
# can only run where /root/reference exists (the GPU box holds bytecode: reference sources are never copied).  These run
# in the CPU suite (no kernel is launched: scripting only).
RUN_ON_CPU = ["nms_rotated::TestScriptable.test_scriptable_cpu"]

_EXPORT = ("TorchScript / tracing EXPORT of the fused ROIPooler / Matcher (ctypes calls into the C ABI) is out of scope "
           "(DESIGN 7: export); ")
_CONTAINER = "exercises only the reference's own container class (Boxes / BoxMode / RotatedBoxes: not a hot-path op)"
NOT_RUN = {
    "nms_rotated::TestScriptable.test_scriptable_cuda": "the same module as test_scriptable_cpu after .cuda() (it has no "
    "parameters); needs the test's source text, which the GPU box does not have -- the _cpu twin runs in the CPU suite",
    "roi_pooler::TestROIPooler.test_scriptability_cpu": _EXPORT + "its eager half is test_roialignv2_roialignrotated_match_*",
    "roi_pooler::TestROIPooler.test_scriptability_gpu": _EXPORT + "its eager half is test_roialignv2_roialignrotated_match_*",
    "roi_pooler::TestROIPooler.test_roi_pooler_tracing": _EXPORT + "shapes per level are covered by tests/test_gpu_pooler.py",
    "matcher::TestMatcher.test_scriptability": _EXPORT + "its known answer (test_matcher.py:16-24) is transcribed in "
    "tests/test_gpu_matcher.py::test_reference_known_answer",
    **{f"rotated_boxes::TestRotatedBoxesStructure.{t}": _CONTAINER for t in (
        "test_clip_area_0_degree", "test_clip_area_arbitrary_angle", "test_normalize_angles", "test_empty_cat",
        "test_scriptability")},
    **{f"boxes::TestBoxMode.{t}": _CONTAINER for t in (
        "test_convert_int_mode", "test_box_convert_list", "test_box_convert_array", "test_box_convert_cpu_tensor",
        "test_box_convert_cuda_tensor", "test_box_convert_xywha_to_xyxy_list", "test_box_convert_xywha_to_xyxy_array",
        "test_box_convert_xywha_to_xyxy_tensor", "test_box_convert_xywh_to_xywha_list",
        "test_box_convert_xywh_to_xywha_array", "test_box_convert_xywh_to_xywha_tensor", "test_json_serializable",
        "test_json_deserializable")},
    **{f"boxes::TestBoxes.{t}": _CONTAINER for t in ("test_empty_cat", "test_to", "test_scriptability")},
}

# torch.jit.script needs the op functions themselves, not the CPU->GPU wrapper: these tests get the product's functions
# bound directly in the test module's namespace and run with the default device set to the GPU (their tensors come from
# torch.rand / random_boxes without an explicit device)
DIRECT = {
    "nms::TestNMS.test_nms_scriptability": ("batched_nms",),
    "nms_rotated::TestScriptable.test_scriptable_cpu": ("nms_rotated",),
}
DEFAULT_DEVICE_GPU = {"nms::TestNMS.test_nms_scriptability"}

_REPORT = {}


def _have():
    return ref.have_py() and all(os.path.exists(build_ref.pyc_path(f)) or ref.have_tree() for f in FILES.values())


def _run(test_id):
    from _reference_surface import Surface

    key, name = test_id.split("::")
    with Surface() as s:
        mod = s.load_test_module(FILES[key])
        for n in DIRECT.get(test_id, ()):
            setattr(mod, n, s.mods["detectron2.layers"].scriptable[n])
        suite = unittest.defaultTestLoader.loadTestsFromName(name, mod)
        stream = io.StringIO()
        prev = torch.get_default_device()
        if test_id in DEFAULT_DEVICE_GPU:
            torch.set_default_device("cuda")
        try:
            res = unittest.TextTestRunner(stream=stream, verbosity=2).run(suite)
        finally:
            torch.set_default_device(prev)
    return res, stream.getvalue()


def test_scripting_only_reference_tests_on_the_cpu():
    """RUN_ON_CPU: needs the reference's source files (TorchScript), no GPU."""
    if not ref.have_tree():
        need_reference(False, "/root/reference (TorchScript needs the test's source text)")
    for test_id in RUN_ON_CPU:
        res, log = _run(test_id)
        assert res.testsRun == 1 and res.wasSuccessful() and not res.skipped, log
        _REPORT[test_id] = "passed (CPU suite)"


def test_lists_cover_exactly_the_reference_tests():
    need_reference(_have(), "the reference's test files (oracle/_ref/py)")
    from _reference_surface import Surface

    found = set()
    with Surface() as s:
        for key, rel in FILES.items():
            mod = s.load_test_module(rel)
            for suite in unittest.defaultTestLoader.loadTestsFromModule(mod):
                for t in suite:
                    found.add(f"{key}::{type(t).__name__}.{t._testMethodName}")
    listed = set(RUN) | set(NOT_RUN) | set(RUN_ON_CPU)
    assert found == listed, (sorted(found - listed), sorted(listed - found))
    assert len(listed) == len(RUN) + len(NOT_RUN) + len(RUN_ON_CPU)


@pytest.mark.gpu
@pytest.mark.parametrize("test_id", RUN)
def test_reference_test(test_id):
    need_reference(_have(), "the reference's test files (oracle/_ref/py)")
    res, log = _run(test_id)
    assert res.testsRun == 1, log
    if res.skipped:
        _REPORT[test_id] = "SKIPPED by the reference's own decorator: " + res.skipped[0][1]
        pytest.fail(f"{test_id} was skipped by the reference itself on a GPU box: {res.skipped[0][1]}")
    ok = res.wasSuccessful()
    _REPORT[test_id] = "passed" if ok else "FAILED"
    assert ok, log + "".join(tb for _, tb in res.failures + res.errors)


@pytest.mark.gpu
def test_report():
    """Prints the executed / not-run reference test ids (after the parametrized cases above: marked `gpu` so that it runs
    in their session)."""
    lines = [f"RUN      {t}: {_REPORT.get(t, 'not executed in this session')}" for t in RUN + RUN_ON_CPU]
    lines += [f"NOT RUN  {t}: {why}" for t, why in sorted(NOT_RUN.items())]
    text = "\n".join(lines)
    print("\n" + text)
    path = os.environ.get("D2AMD_REFERENCE_TEST_REPORT")
    if path:
        with open(path, "w") as f:
            f.write(text + "\n")
