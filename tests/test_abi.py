"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/alignn_hip.h declares (no compute calls - there is no GPU here), and the ctypes table in
alignn_amd/_lib.py covers exactly that set."""

import os
import re

from alignn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "alignn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(alignn_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from alignn_amd.build import build

    build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert lib.alignn_version().decode().startswith("alignn_hip")


def test_ctypes_table_matches_header():
    assert set(_lib.SIGNATURES) == _declared()


def test_host_only_queries():
    lib = _lib.load()
    assert lib.alignn_col_stats_slabs(1) == 1
    assert lib.alignn_col_stats_slabs(10**7) == 1024 and lib.alignn_col_stats_slabs(3840) == 120
    assert lib.alignn_egc_slabs(0) == 1
    ws = lib.alignn_gemm_tn_workspace(10000, 256, 256)
    assert ws % (256 * 256 * 4) == 0 and 1 <= ws // (256 * 256 * 4) <= -(-10000 // 128)
