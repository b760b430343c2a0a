"""CPU-side checks of the drop-in boundary: libbie_hip.so loads without a GPU, exports every symbol that
include/bie_hip.h declares, validates arguments before touching the device, and its pure-host entry points agree
with the oracle.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bie_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bie_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from bitorch_engine import _hip
    lib = _hip.lib()
    syms = declared_symbols()
    assert len(syms) >= 22
    for s in syms:
        assert hasattr(lib, s), f"libbie_hip.so does not export {s}"
    assert sorted(_hip.SIGNATURES) == syms, "ctypes signature table out of sync with include/bie_hip.h"
    assert lib.bie_version() == 300


def test_argument_validation_happens_before_any_device_work():
    from bitorch_engine import _hip
    L = _hip.lib()
    assert L.bie_mpq_forward(None, None, None, None, None, None, None, None, 0, 1, 64, 32, 5, 32, 0, 0, None) == -2
    assert b"w_bit=5" in L.bie_last_error()
    assert L.bie_mpq_forward(None, None, None, None, None, None, None, None, 0, 1, 64, 32, 4, 32, 0, 0, None) == -1
    assert L.bie_mpq_dequant(None, None, None, None, None, 60, 32, 4, 32, 0, 0, None) == -1  # K % 8 != 0
    assert L.bie_mbwq_q4_dequant(None, None, None, None, None, 64, 32, 3, 32, None) == -1
    assert L.bie_binary_linear_forward(None, None, None, 1, 8, 60, 0, 1.0, None) == -1
    assert L.bie_mpq_workspace_bytes(1, 4096, 11008, 4) > 0
    assert L.bie_mpq_workspace_bytes(4096, 4096, 11008, 4) >= 0
    assert L.bie_mpq_workspace_bytes(0, 4096, 11008, 4) == 0
    assert L.bie_mpq_forward_grouped(None, 9, None, None, None, None, None, None, None, 0, 1, 64, 4, 32, 0, 1, None) == -1
    assert L.bie_mpq_forward_grouped(None, 2, None, None, None, None, None, None, None, 0, 1, 64, 4, 32, 0, 1, None) == -1
    assert L.bie_workspace_init(None, 0, None) == -3
    assert L.bie_mpq_sort_rows(None, None, None, 64, 32, 3, None) == -2
    assert L.bie_mpq_sort_rows(None, None, None, 60, 32, 4, None) == -1
    assert L.bie_gather_cols(None, None, None, 0, 64, 1, None) == 0  # empty batch
    assert L.bie_gather_cols(None, None, None, 2, 64, 1, None) == -1
    n2 = (ctypes.c_int * 2)(4096, 11008)
    assert L.bie_mpq_grouped_workspace_bytes(2, n2, 1, 4096, 4) >= L.bie_mpq_workspace_bytes(1, 4096, 11008, 4)
    assert L.bie_binary_conv2d_workspace_bytes(32, 512, 7, 7, 512, 3, 1, 1, 1) == 16384 + (32 * 49 * 144 + 512 * 144) * 4  # behind the counter head


def test_mbwq_rows_host_function_matches_oracle_and_reference_tables(golden_dir):
    from bitorch_engine import _hip
    L = _hip.lib()
    g = np.load(os.path.join(golden_dir, "exl2_group_maps.npz"))
    for c in ("q_proj", "k_proj", "w3w2", "all6"):
        K = int(g[c + "_meta"][0])
        qg = np.ascontiguousarray(g[c + "_q_groups"])
        rows = (ctypes.c_int * 7)()
        assert L.bie_mbwq_rows(qg.ctypes.data, qg.size // 2, K, ctypes.cast(rows, ctypes.c_void_p)) == 0
        assert list(rows) == orc.exl2_rows(qg, K)
    bad = np.array([7, 0], np.int16)
    rows = (ctypes.c_int * 7)()
    assert L.bie_mbwq_rows(bad.ctypes.data, 1, 32, ctypes.cast(rows, ctypes.c_void_p)) == -2


def test_missing_library_fails_loudly(monkeypatch):
    from bitorch_engine import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libbie_hip.so")
    with pytest.raises(RuntimeError, match="not found"):
        _hip.lib()
