"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/slim_hip.h
declares, the ctypes table mirrors the header, and the host-only entry points work."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "slim_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slim_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_functions():
    names = declared_functions()
    assert "slim_mask_nm" in names and "slim_lowrank_fit" in names and len(names) >= 30


def test_library_exports_every_declared_symbol():
    from slim_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in slim_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == declared_functions()
    assert lib.slim_abi_version() == 1


def test_errors_are_status_codes_not_exceptions():
    from slim_amd import _lib
    lib = _lib.load()
    # null pointers are rejected before any GPU work is attempted
    rc = lib.slim_mask_nm(None, None, 4, 8, 2, 4, None, None)
    assert rc == _lib.ERR_INVALID and b"slim_mask_nm" in lib.slim_last_error()
    assert lib.slim_quant_search_workspace_bytes(512) == (3 * 512 + 32 + 256 + 1024 + 512 + 1024) * 4      # + the lists of occupied bins
    assert lib.slim_lowrank_fit_workspace_bytes(1024, 512, 192) > 4 * (1024 + 512) * 192 * 2
    with pytest.raises(ValueError):
        _lib.call("slim_shift_zeros", None, None, 0, None)


def test_tie_table_matches_reference(golden):
    """The 4-wide selection table (host code, no GPU) against all weak orderings run through the reference."""
    from slim_amd import _lib
    lib = _lib.load()
    g = golden("g3_prune_nm")
    tups = g["tuples"].numpy().astype(int)
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    for n in (1, 2, 3):
        lut = (ctypes.c_uint8 * 729)()
        assert lib.slim_nm_lut4(n, lut) == 0
        ref = g[f"mask_n{n}"].numpy()
        for t, m in zip(tups, ref):
            sig = sum((0 if t[a] < t[b] else 1 if t[a] == t[b] else 2) * 3 ** p for p, (a, b) in enumerate(pairs))
            bits = [(lut[sig] >> i) & 1 for i in range(4)]
            assert bits == m.tolist(), (n, t, bits, m)


def test_ops_refuse_cpu_tensors():
    from slim_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.mask_nm(torch.zeros(4, 8, dtype=torch.bfloat16), None, 2, 4)
