"""CPU tests of the C-ABI boundary: the shared library loads without a GPU and exports
exactly the entry points ``include/pgmuvi_hip.h`` declares (no compute call is made)."""
import ctypes
import os
import re

import pytest

from pgmuvi_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pgmuvi_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("pgm_workspace_create", "pgm_workspace_destroy", "pgm_sm_kernel_f64", "pgm_mll_value_grad_f64",
                 "pgm_mll_value_grad_batched_f64", "pgm_predict_f64"):
        assert must in names
    txt = open(HEADER).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and 'extern "C"' in code     # plain C ABI, no torch types
    for cite in ("pgmuvi/trainers.py:179-181", "pgmuvi/gps.py:208", "pgmuvi/lightcurve.py:9607"):
        assert cite in txt                                           # each entry point cites what it replaces


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.lib_path())
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_hip.SYMBOLS) == _declared()
    lib = _hip.load()
    assert _hip.version().startswith("pgmuvi_hip") and "gfx950" in _hip.version()
    assert _hip.max_qd() == 16
    assert lib.pgm_profile_phases() == 8
    names = [lib.pgm_profile_phase_name(i).decode() for i in range(7)]
    assert "trailing_update" in names and "diag_block" in names


def test_code_object_is_gfx950_only():
    blob = open(_hip.lib_path(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob
