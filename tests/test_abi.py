"""The C-ABI shared library must load on a machine WITHOUT a GPU and export every symbol
include/epievo_mi355x.h declares; the product has no CPU fallback, so creating a context
here must fail cleanly rather than route elsewhere."""
import ctypes
import os
import re

import pytest

from epievo_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "epievo_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epv_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_build.HIP_SO), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_build.HIP_SO)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    from epievo_amd.sampler import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from epievo_amd.sampler import DeviceSampler
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        DeviceSampler(0)


def test_product_does_not_import_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "epievo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")) and f != "_build.py":
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("liborc", "import orc", "from orc", '#include "orc', "libepievo_ref", "dlopen"):
                    assert needle not in text, (f, needle)
