"""The C-ABI library loads and exports every function include/aig_abi.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT
from gan_amd import _abi


def declared_functions():
    text = open(os.path.join(ROOT, "include", "aig_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aig_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 10
    assert os.path.exists(_abi.LIB_PATH), "libaig_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_abi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    assert sorted(_abi.SIGNATURES) == declared_functions()


def test_version_and_error_string_callable_without_gpu():
    lib = _abi.lib()
    assert lib.aig_abi_version() >= 1
    assert isinstance(lib.aig_last_error(), bytes)
