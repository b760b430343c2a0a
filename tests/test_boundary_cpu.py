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


def declared_symbols(header="bie_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
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
    assert not [s for s in syms if "test" in s], "fault-injection hooks do not belong in the drop-in ABI header"
    hooks = declared_symbols("bie_hip_testing.h")
    assert sorted(_hip.TEST_HOOKS) == hooks and all(hasattr(lib, s) for s in hooks)
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


def test_exl2_grouped_and_lone_workspace_sizes_are_host_functions_of_the_tables():
    """bie_mbwq_exl2_grouped_workspace_bytes / bie_mbwq_workspace_bytes (host only: they read K, N and the host band tables, never a tensor):
    a group is sized for 1 ... 16 rows of x (its x block and chunk sums are written for the instantiated row count 1 / 2 / 4 / 8 / 16), refuses 17
    rows, an irregular member and more than 8 members; a lone call's workspace covers the group-of-one path it may take for every row count."""
    from bitorch_engine import _hip
    L = _hip.lib()

    def table(spec):
        qg, row, K = [], 0, 0
        for bits, k in spec:
            qg += [bits, row]
            row += k * bits // 32
            K += k
        qg = np.array(qg, np.int16)
        t = (ctypes.c_int * 20)()
        assert L.bie_mbwq_exl2_table(qg.ctypes.data, qg.size // 2, K, ctypes.cast(t, ctypes.c_void_p)) == 0
        return t, K
    reg, K = table([(4, 128)] * 8 + [(3, 32)] * 48 + [(2, 64)] * 24)
    odd, K2 = table([(4, 96)] * 32 + [(2, 32)] * 32)
    assert K == K2 == 4096 and (reg[6] & 0x200) and not (odd[6] & 0x200)
    fake = 0x10000  # tensor pointers are not read by the size functions
    ent = lambda t, N: _hip.Exl2ListEntry(fake, fake, fake, fake, fake, fake, ctypes.cast(t, ctypes.c_void_p), fake, K, N, 0, 0)
    two = (_hip.Exl2ListEntry * 2)(ent(reg, 4096), ent(reg, 11008))
    sizes = [L.bie_mbwq_exl2_grouped_workspace_bytes(2, two, M) for M in (1, 2, 3, 4, 5, 8, 9, 16)]
    assert all(s > 16384 for s in sizes) and sizes == sorted(sizes)
    assert sizes[2] == sizes[3] and sizes[4] == sizes[5] and sizes[6] == sizes[7]   # 3 -> 4, 5 -> 8, 9 -> 16 rows instantiated
    xrows = lambda M: 2 * (M * K * 2 + M * (K // 32) * 8)                            # both members' x block + chunk sums
    assert sizes[7] - sizes[5] >= xrows(16) - xrows(8)
    assert L.bie_mbwq_exl2_grouped_workspace_bytes(2, two, 17) == 0 and L.bie_mbwq_exl2_grouped_workspace_bytes(2, two, 0) == 0
    mixed = (_hip.Exl2ListEntry * 2)(ent(reg, 4096), ent(odd, 4096))
    assert L.bie_mbwq_exl2_grouped_workspace_bytes(2, mixed, 1) == 0
    nine = (_hip.Exl2ListEntry * 9)(*[ent(reg, 256)] * 9)
    assert L.bie_mbwq_exl2_grouped_workspace_bytes(9, nine, 1) == 0 and L.bie_mbwq_exl2_grouped_workspace_bytes(8, nine, 1) > 0
    one = (_hip.Exl2ListEntry * 1)(ent(reg, 4096))
    for M in (1, 2, 3, 8, 16):
        assert L.bie_mbwq_workspace_bytes(M, K, 4096) >= L.bie_mbwq_exl2_grouped_workspace_bytes(1, one, M) - 16384, M


def test_exl2_table_host_function_marks_regular_group_structures_and_their_shifts(golden_dir):
    """bie_mbwq_exl2_table (host only; what bie_mbwq_exl2_shuffle returns beside re-arranging the tensor): the reference's 7 ints first, the
    SHUFFLED mark, REGULAR exactly when every band's groups hold the same power-of-two number of whole 32-k chunks (a shorter last group
    allowed), and per band the first group and log2(chunks per group) the direct decode form derives a chunk's group from -- checked
    against the group map the reference builds (make_group_map) for every chunk."""
    import torch
    from bitorch_engine import _hip
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    L = _hip.lib()
    g = np.load(os.path.join(golden_dir, "exl2_group_maps.npz"))
    cases = [(np.ascontiguousarray(g[c + "_q_groups"]), int(g[c + "_meta"][0]), True) for c in ("q_proj", "k_proj", "w3w2", "all6")]
    for spec, regular in (([(4, 64)] * 3 + [(4, 32)] + [(3, 128)] * 2 + [(3, 96)] + [(2, 64)] * 4, True), ([(6, 96), (4, 160), (2, 224)], True),
                          ([(4, 32), (4, 64), (4, 32)] + [(2, 64)] * 4, False), ([(4, 16)] * 8 + [(2, 16)] * 12, False), ([(4, 96)] * 3 + [(2, 32)] * 4, False)):
        qg, row, K = [], 0, 0
        for bits, k in spec:
            qg += [bits, row]
            row += k * bits // 32
            K += k
        cases.append((np.array(qg, np.int16), K, regular))
    for qg, K, regular in cases:
        groups = qg.size // 2
        t = (ctypes.c_int * 20)()
        assert L.bie_mbwq_exl2_table(qg.ctypes.data, groups, K, ctypes.cast(t, ctypes.c_void_p)) == 0
        t = list(t)
        ref = orc.exl2_rows(qg, K)
        assert t[:6] == ref[:6] and (t[6] & 0xff) == ref[6] and (t[6] & 0x100) and t[19] == 0x45584c32
        assert bool(t[6] & 0x200) == regular, (qg.tolist(), hex(t[6]))
        if not regular:
            continue
        rows_packed = int(sum(((int(qg[2 * i + 3]) - int(qg[2 * i + 1])) if i < groups - 1 else 0) for i in range(groups)))
        bits_l = qg[0::2].astype(int)
        k_per = [(int(qg[2 * i + 3]) - int(qg[2 * i + 1])) * 32 // int(bits_l[i]) for i in range(groups - 1)]
        k_per.append(K - sum(k_per))
        rows_packed += k_per[-1] * int(bits_l[-1]) // 32
        gmap = make_group_map(torch.from_numpy(qg), rows_packed).numpy()  # (group, rows left) per k
        band_end, kprev = t[:6], 0
        for b in range(6):
            for c in range(kprev // 32, band_end[b] // 32):
                want = int(gmap[2 * (32 * c)])
                got = t[7 + b] + ((c - kprev // 32) >> t[13 + b])
                assert got == want, (b, c, got, want)
            kprev = band_end[b]


def test_missing_library_fails_loudly(monkeypatch):
    from bitorch_engine import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libbie_hip.so")
    with pytest.raises(RuntimeError, match="not found"):
        _hip.lib()


def test_every_extension_function_takes_the_reference_positional_argument_list(golden_dir):
    """tests/golden/extension_signatures.json = name -> positional parameter list of every function the reference binds with
    pybind11 (m.def without py::arg: positional only), read off the reference's C++ by oracle/gen_golden.py.  Every one of them
    must exist in the same-named module here and accept exactly that many positional arguments (extra parameters only with
    defaults) -- a caller written against the reference's extension modules must not be able to hit a silently different meaning
    (round 2: binary_linear_cutlass.mm took `transpose` where the reference takes `kernel_id`)."""
    import importlib
    import inspect
    import json
    sig = json.load(open(os.path.join(golden_dir, "extension_signatures.json")))
    assert len(sig) == 9 and sum(len(v) for v in sig.values()) == 36
    for mod_name, fns in sig.items():
        mod = importlib.import_module("bitorch_engine.extensions." + mod_name)
        for name, params in fns.items():
            assert hasattr(mod, name), f"{mod_name}.{name} is missing"
            ps = list(inspect.signature(getattr(mod, name)).parameters.values())
            assert all(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in ps), f"{mod_name}.{name}: non-positional parameter"
            required = [p for p in ps if p.default is inspect.Parameter.empty]
            assert len(required) <= len(params) <= len(ps), f"{mod_name}.{name} takes {[p.name for p in ps]}, the reference {params}"
    # the parameter whose MEANING was wrong in round 2
    from bitorch_engine.extensions import binary_linear_cutlass
    assert list(inspect.signature(binary_linear_cutlass.mm).parameters) == ["x", "y", "kernel_id"]


def test_list_plan_validation_without_a_device():
    """bie_mpq_list_*: sizes and argument errors are decided on the host before any device work."""
    from bitorch_engine import _hip
    L = _hip.lib()
    E = _hip.ListEntry
    ok = (E * 2)(E(8, 8, 8, 8, None, 8, 4096, 4096, -1, 0), E(8, 8, 8, 8, None, 8, 4096, 11008, -1, 0))
    n = L.bie_mpq_list_device_bytes(2, ok, 1, 4, 128)
    assert n >= 2 * 128 + (64 + 172) * 8
    assert L.bie_mpq_list_device_bytes(2, ok, 3, 4, 128) > 0   # 3 <= M <= 32: the matrix-pipe list kernel (W4 only)
    assert L.bie_mpq_list_device_bytes(2, ok, 17, 4, 128) > 0  # 17 .. 32: two row blocks per pass
    assert L.bie_mpq_list_device_bytes(2, ok, 33, 4, 128) == 0  # M > 32
    assert L.bie_mpq_list_device_bytes(2, ok, 3, 2, 128) == 0   # W2: M <= 2
    assert L.bie_mpq_list_device_bytes(2, ok, 1, 8, 128) == 0  # w_bit 8
    bad = (E * 1)(E(8, 8, 8, 8, None, 8, 4000, 64, -1, 0))
    assert L.bie_mpq_list_device_bytes(1, bad, 1, 4, 128) == 0  # K not a multiple of the group
    h = ctypes.c_void_p()
    dep = (E * 2)(E(8, 8, 8, 8, None, 16, 4096, 4096, 1, 0), E(16, 8, 8, 8, None, 24, 4096, 4096, -1, 0))
    assert L.bie_mpq_list_create(ctypes.byref(h), 2, dep, 1, 4, 128, 0, 1, 256, 1 << 30) == -1
    assert b"not EARLIER" in L.bie_last_error()
    dep2 = (E * 2)(E(8, 8, 8, 8, None, 16, 4096, 4096, -1, 0), E(32, 8, 8, 8, None, 24, 4096, 4096, 0, 0))
    assert L.bie_mpq_list_create(ctypes.byref(h), 2, dep2, 1, 4, 128, 0, 1, 256, 1 << 30) == -1
    assert b"does not read its y" in L.bie_last_error()
    assert L.bie_mpq_list_create(ctypes.byref(h), 2, ok, 1, 4, 128, 0, 1, 256, 16) == -3  # device buffer too small
    assert L.bie_mpq_list_forward(None, None) == -1
    assert L.bie_device_status(0) == 0  # no status page without a GPU: reads as clear


def test_the_512_register_gemm_kernels_do_not_spill():
    """The 256 x 256 dense GEMM instances run one wave per SIMD on all 512 registers (256 accumulators): a change that costs them a few
    more is compiled WITH scratch -- still correct, 5-10 % slower, and silent (round 5 hit it three times while editing the epilogue).
    hipcc's own resource report (-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU) must say ScratchSize 0 for every
    mpq_dense_gemm_kernel instance."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "bitorch-engine_amd", "csrc", "mpq_dense.hip")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
                        "--cuda-device-only", "-c", src, "-o", os.devnull], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    name, seen = None, {}
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and "mpq_dense_gemm_kernel" in name:
            seen[name] = int(m.group(1))
    assert len(seen) >= 4, f"resource report not parsed: {list(seen)}"
    assert all(v == 0 for v in seen.values()), f"a dense GEMM instance spills: {seen}"


def test_workspace_sizing_is_total_over_row_counts_and_shapes():
    """The host-only sizing functions answer for EVERY (M, K, N), also where a plan is not defined for M: bie_mbwq_workspace_bytes asked the
    one/two-row decode plan about M >= 769 on shapes the prefill form cannot take (N % 8 != 0, or the prefill switched off) and divided by
    zero -- a SIGFPE in the caller's process (found by the knob matrix, tools/gpu_knob_matrix.sh).  Monotone enough to be usable: more rows
    never need less than one row."""
    from bitorch_engine import _hip
    L = _hip.lib()
    for (K, N) in ((4096, 4096), (4096, 11008), (4096, 100), (1024, 12), (11008, 4100), (32, 8)):
        one = L.bie_mbwq_workspace_bytes(1, K, N)
        assert one > 0
        for M in (2, 3, 16, 17, 48, 49, 64, 65, 768, 769, 1000, 4096, 65536):
            b = L.bie_mbwq_workspace_bytes(M, K, N)
            assert b > 0, (M, K, N)
    for w_bit in (1, 2, 4, 8):
        for (K, N) in ((4096, 4096), (4096, 100), (11008, 4100)):
            for M in (1, 2, 16, 17, 32, 33, 1000, 4096, 65536):
                assert L.bie_mpq_workspace_bytes(M, K, N, w_bit) > 0, (M, K, N, w_bit)


def test_prefill_dispatch_follows_the_measured_table_inside_its_grid(monkeypatch):
    """mpq_dense_ok (csrc/mpq_dense.hip): inside the measured grid the dense two-launch form is chosen where the sweep found it faster
    (profiles/r06_dense_rule_sweep.txt -> csrc/mpq_dense_table.inc), by the nearest grid point in log space; outside the grid and with
    BIE_GEMM_DENSE_TABLE=0 the round-3 analytic rule answers.  Seen through bie_mpq_prefill_form (host-only); the sizing function must agree: a plan that takes the dense form holds the
    K x N x 2-byte fragment image.  (BIE_TUNING, set by tests/conftest.py, makes the library re-read its knobs per call.)"""
    from bitorch_engine import _hip
    L = _hip.lib()
    dense = lambda M, K, N: L.bie_mpq_prefill_form(M, K, N) == 1
    for (M, K, N, want) in ((1024, 4096, 4096, True), (1024, 4096, 11008, False), (4096, 4096, 11008, True), (1024, 14336, 4096, False), (3072, 8192, 8192, True),
                            (1000, 4000, 4104, True),      # off the grid: the nearest point is (1024, 4096, 4096)
                            (896, 4096, 4096, False),      # below the row floor
                            (512, 4096, 4096, False)):
        assert dense(M, K, N) == want, (M, K, N)
    for (M, K, N) in ((1024, 4096, 4096), (4096, 4096, 11008), (3072, 8192, 8192)):
        assert L.bie_mpq_workspace_bytes(M, K, N, 4) >= K * N * 2
    assert dense(8192, 4096, 4096)             # beyond the row grid: the analytic rule (whole rounds of 256 x 256 tiles, short K)
    assert not dense(2048, 1024, 1024)         # below the K / N grid: the analytic rule (too few tiles)
    monkeypatch.setenv("BIE_GEMM_DENSE_TABLE", "0")  # BIE_TUNING (tests/conftest.py): knobs are re-read per call
    assert not dense(3072, 8192, 8192), "without the table the round-3 rule keeps K = 8192 fused below 4096 rows"
    assert dense(1024, 4096, 4096)


def test_rows_form_follows_the_measured_decode_table():
    """bie_mpq_rows_form (host-only): the decode kernels take every W4 call up to 16 rows, 17 .. 32 rows only on the measured shapes and up to the measured row count per
    dtype (csrc/mpq_lut_rb2_table.inc), never W2 beyond two rows; prefill rows report the dense / fused choice of bie_mpq_prefill_form."""
    from bitorch_engine import _hip
    L = _hip.lib()
    F16, BF16 = _hip.F16, _hip.BF16
    for (M, K, N, w, dt, want) in ((1, 4096, 4096, 4, BF16, 2), (16, 4096, 2048, 4, F16, 2), (17, 4096, 4096, 4, F16, 2), (32, 4096, 4096, 4, BF16, 2), (33, 4096, 4096, 4, F16, 0),
                                   (24, 4096, 11008, 4, F16, 2), (24, 4096, 11008, 4, BF16, 0), (32, 4096, 2048, 4, F16, 0), (20, 11008, 4096, 4, F16, 2), (24, 11008, 4096, 4, F16, 0),
                                   (17, 4096, 4160, 4, F16, 0), (17, 4096, 4096, 2, F16, 0), (2, 4096, 4096, 2, F16, 2), (1024, 4096, 4096, 4, BF16, 1), (512, 4096, 4096, 4, BF16, 0)):
        assert L.bie_mpq_rows_form(M, K, N, w, 128, dt) == want, (M, K, N, w, dt)
    # the workspace the library asks for covers the decode kernel's partial sums at 17 .. 32 rows (it used to stop at 16: a fault on the first sweep of this form)
    assert L.bie_mpq_workspace_bytes(28, 4096, 6144, 4) >= L.bie_mpq_workspace_bytes(16, 4096, 6144, 4)
