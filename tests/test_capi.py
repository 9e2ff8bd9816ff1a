"""not-gpu: the C-ABI library loads on a GPU-less host and exports every symbol that
include/uda_clr_hip.h declares (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "uda_clr_amd", "lib", "libuda_clr_hip.so")


def _declared():
    txt = open(os.path.join(ROOT, "include", "uda_clr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(uda_[a-z0-9_]+)\s*\(", txt)))


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    from uda_clr_amd.kernels import SYMBOLS, load_library
    lib = load_library()
    declared = _declared()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert set(declared) == set(SYMBOLS), (set(declared) ^ set(SYMBOLS))
    assert lib.uda_version() >= 1
    assert lib.uda_last_error() is not None


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "uda_clr_hip.h")).read()
    for anchor in ("mobilenet.py", "aspp.py", "decoder.py", "Trainer_prototype_full.py", "utils/Utils.py", "utils/metrics.py"):
        assert anchor in txt
