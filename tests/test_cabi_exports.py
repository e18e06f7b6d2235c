"""CPU-only: the C-ABI library loads and exports every symbol include/g2v.h declares (no compute calls)."""
import ctypes
import os
import re

from gesture2vec_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "g2v.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(g2v_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(_lib.EXPORTS) == decl


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} not exported by libg2v_hip.so"
    assert b"g2v" in lib.g2v_version()


def test_argument_errors_are_reported_not_thrown():
    lib = _lib.load()
    rc = lib.g2v_vq_code_sqnorm(None, None, 4, 4, None)
    assert rc == -1
    assert b"null" in lib.g2v_last_error()
    assert lib.g2v_vq_assign_blocks(4096) == 256
    assert lib.g2v_linear_bwd_weight_workspace(1000, 64, 192) > 0
