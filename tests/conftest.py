import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# GPU run order (the driver uses -x): the core parity files of SURVEY 8(a) first, then the fused 8(f) rows, the DCN
# configuration sweep last -- a failure late in the list must not hide the core results.
_ORDER = ["test_gpu_parity", "test_gpu_pooler", "test_gpu_masks", "test_gpu_matcher", "test_gpu_mask_head",
          "test_gpu_rpn", "test_gpu_dense", "test_gpu_nms_scale", "test_gpu_polygons", "test_gpu_cshim",
          "test_gpu_dcn_tc"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else (-1 if not name.startswith("test_gpu") else len(_ORDER) - 1)

    items.sort(key=rank)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
