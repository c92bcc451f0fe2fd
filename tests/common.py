"""Shared helpers for the test-suite (inputs, configurations): the workloads live in the
package (epievo_amd/workloads.py) so that bench.py does not depend on the test tree."""
import os

from epievo_amd.workloads import (EXTRA_TREES, TEST_PARAM_TEXT, TREE_NWK_TEXT, _tmp, config,  # noqa: F401
                                  ref_test_model, simulate, tree_nwk)

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
