import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the native pieces are built in-tree by __graft_entry__.build(); make sure they exist
    # (and are current) wherever the suite runs -- hipcc cross-compiles without a GPU
    from epievo_amd import _build
    _build.build_all()


@pytest.fixture(scope="session")
def ref_test_model():
    """test/test.param of the reference (stationary 0.844912 0.893359 / baseline -0.8 -1.8)
    scaled to unit rate; values are the reference's own (golden/model.json pins them)."""
    from common import ref_test_model as tm
    return tm()


@pytest.fixture(scope="session")
def tree_nwk():
    from common import tree_nwk as t
    return t()
