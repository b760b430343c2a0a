#!/usr/bin/env python3
"""Randomised shape sweeps of the remaining operators against the CPU restatement (bit-exact except the uniform MBWQ forward): binary linear
   (XNOR kernels and the matrix-pipe form on either side of its switch), binary conv (tap form / im2col form / matrix-pipe form by geometry), the
   W4A4 / W8A8 integer GEMMs, uniform MBWQ q4 / q2 (dequantised weight bit-exact, forward within the parity gate), grouped decode calls.
   A refusal (RuntimeError) is fine; a wrong value, a NaN or a crash is a finding.
   usage: python tests/sweeps/fuzz_other_ops.py [cases=60 per operator] [seed=1]   (test infrastructure: imports oracle/)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = T.DEV


def run(cases=60, seed=1, ops=("binlin", "binconv", "binconv1", "q4", "q8", "mbwq", "grouped", "grad", "pack")):
    from bitorch_engine.extensions import binary_linear_cuda, q_linear_cutlass as qc, q_linear_cuda
    from bitorch_engine.extensions._binary_common import pack_rows, conv2d
    rng = np.random.default_rng(seed)
    res = {}
    for op in ops:
        refused, ok, bad = {}, 0, []
        for c in range(cases):
            gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
            tag = op
            try:
                if op == "binlin":
                    M = int(rng.choice([1, 2, 3, 5, 8, 31, 64, 65, 128, 191, 192, 193, 256, 300, 700]))
                    N = int(rng.choice([1, 7, 8, 40, 63, 64, 65, 128, 200, 520, 1000]))
                    K = 8 * int(rng.integers(1, 260)) if rng.random() < 0.5 else 128 * int(rng.integers(1, 17))
                    tag = f"binlin M={M} N={N} K={K}"
                    x, w = torch.randn((M, K), generator=gen), torch.randn((N, K), generator=gen)
                    y = binary_linear_cuda.forward(x.to(DEV), w.to(DEV), 3, True).cpu().numpy()
                    ref = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.numpy()), orc.binary_pack_rows(w.numpy()), K)
                    good = np.array_equal(y, ref)
                elif op == "binconv":
                    B, C = int(rng.integers(1, 5)), int(rng.choice([3, 8, 16, 24, 32, 40, 64, 96, 128, 256]))
                    H, W = int(rng.integers(3, 30)), int(rng.integers(3, 30))
                    OC = int(rng.choice([1, 4, 8, 36, 64, 70, 128, 130]))
                    ks = int(rng.choice([1, 2, 3, 3, 5]))
                    st, pad, dil = int(rng.choice([1, 1, 2])), int(rng.integers(0, 3)), int(rng.choice([1, 1, 2]))
                    if (C * ks * ks) % 8 or H + 2 * pad < dil * (ks - 1) + 1 or W + 2 * pad < dil * (ks - 1) + 1:
                        continue
                    tag = f"binconv B={B} C={C} H={H} W={W} OC={OC} k={ks} s={st} p={pad} d={dil}"
                    x, w = torch.randn((B, C, H, W), generator=gen), torch.randn((OC, C, ks, ks), generator=gen)
                    wp = pack_rows(w.reshape(OC, -1).to(DEV)).contiguous()
                    y = conv2d(x.to(DEV), wp, OC, ks, st, pad, dil, 1.0).cpu().numpy()
                    good = np.array_equal(y, orc.binary_conv2d(x.numpy(), w.numpy(), st, pad, dil))
                elif op == "binconv1":
                    # the ONE-launch forms of round 6 (binary_conv_fused.hip) through the C ABI, both of them where the geometry allows, on geometries drawn
                    # inside their range: every channel count / kernel size they take, strides, paddings, ragged channel blocks, all three input dtypes
                    from bitorch_engine import _hip
                    from bitorch_engine.extensions import _binary_common as bc
                    L = _hip.lib()
                    B, C = int(rng.integers(1, 12)), int(rng.choice([64, 128, 256, 512]))
                    H, W = int(rng.integers(2, 30)), int(rng.integers(2, 34))
                    OC = int(rng.choice([1, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 300]))
                    ks, st, pad = int(rng.choice([1, 3, 3])), int(rng.choice([1, 1, 2, 3])), int(rng.integers(0, 3))
                    if H + 2 * pad < ks or W + 2 * pad < ks:
                        continue
                    dts = str(rng.choice(["f32", "f16", "bf16"]))
                    tag = f"binconv1 {dts} B={B} C={C} H={H} W={W} OC={OC} k={ks} s={st} p={pad}"
                    x, w = torch.randn((B, C, H, W), generator=gen), torch.randn((OC, C, ks, ks), generator=gen)
                    x.view(-1)[::53] = 0.0
                    x.view(-1)[7::97] = -0.0
                    xd = x.to(T._TDT[dts]).to(DEV)
                    wp = pack_rows(w.reshape(OC, -1).to(DEV)).contiguous()
                    want = orc.binary_conv2d(xd.float().cpu().numpy(), w.numpy(), st, pad, 1)
                    OH, OW = want.shape[2], want.shape[3]
                    good, ran = True, 0
                    stream = torch.cuda.current_stream().cuda_stream
                    if L.bie_binary_conv2d_fused_ok(B, C, H, W, OC, ks, st, pad, 1):
                        y = torch.full((B, OC, OH, OW), float("nan"), device=DEV)
                        rc = L.bie_binary_conv2d_forward_fused(xd.data_ptr(), bc.conv_weight_lanes(wp, OC, C, ks).data_ptr(), y.data_ptr(), B, C, H, W, OC, ks, st, pad, 1, 1.0,
                                                               _hip.dt(xd), stream)
                        good, ran = good and rc == 0 and np.array_equal(y.cpu().numpy(), want), ran + 1
                    if L.bie_binary_conv2d_mfma_ok(B, C, H, W, OC, ks, st, pad, 1):
                        y = torch.full((B, OC, OH, OW), float("nan"), device=DEV)
                        rc = L.bie_binary_conv2d_forward_mfma(xd.data_ptr(), bc.conv_weight_fp4_image(wp, OC, C, ks).data_ptr(), y.data_ptr(), B, C, H, W, OC, ks, st, pad, 1, 1.0,
                                                              _hip.dt(xd), stream)
                        good, ran = good and rc == 0 and np.array_equal(y.cpu().numpy(), want), ran + 1
                    if ran == 0:
                        continue
                elif op == "q4":
                    M = int(rng.choice([1, 5, 64, 127, 128, 129, 130, 257, 300, 512]))
                    N = int(rng.choice([4, 36, 124, 128, 132, 260, 520, 1024]))
                    K = 64 * int(rng.integers(1, 40))
                    dts = str(rng.choice(["f16", "bf16", "f32"]))
                    tag = f"q4 {dts} M={M} N={N} K={K}"
                    tdt = T._TDT[dts]
                    x = torch.randn((M, K), generator=gen).to(tdt)
                    w = (torch.randn((N, K), generator=gen) * 0.05).to(tdt)
                    sa, sw = float(2 * x.float().abs().mean() / 11.269), float(2 * w.float().abs().mean() / 5.6345)
                    code = orc.dt_code(tdt)
                    want = orc.q4_gemm(orc.q4_quantize_pack(orc.torch_to_np(x), sa, code), orc.q4_quantize_pack(orc.torch_to_np(w), sw, code), K, sa, sw, code)
                    out = qc.q4_forward(x.to(DEV), w.to(DEV), torch.tensor(sa), torch.tensor(sw), False, False)[0]
                    good = np.array_equal(orc.torch_to_np(out), want)
                elif op == "q8":
                    M = int(rng.choice([1, 33, 127, 128, 129, 256, 300, 512]))
                    N = int(rng.choice([4, 100, 128, 132, 384, 520, 1024]))
                    K = 64 * int(rng.integers(1, 40))
                    tag = f"q8 M={M} N={N} K={K}"
                    a = rng.integers(-128, 128, (M, K)).astype(np.int8)
                    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
                    got = qc.q8_forward(torch.from_numpy(a).to(DEV), torch.from_numpy(w).to(DEV), False, torch.tensor(0.013), torch.tensor(0.0021))
                    good = np.array_equal(got.cpu().numpy(), orc.q8_gemm(a, w, 0.013, 0.0021))
                elif op == "mbwq":
                    bits = int(rng.choice([2, 4]))
                    gs = int(rng.choice([32, 64, 128]))
                    K = gs * int(rng.integers(1, 1024 // gs + 1))
                    u = rng.random()
                    N = 8 * int(rng.integers(1, 80)) if u < 0.5 else (4 * int(rng.integers(1, 100)) if u < 0.8 else int(rng.integers(1, 300)))
                    M = int(rng.choice([1, 2, 3, 8, 16, 17, 33, 40, 64, 100, 300, 1100]))
                    perm = bool(rng.random() < 0.5)
                    tag = f"mbwq q{bits} g{gs} K={K} N={N} M={M} perm={perm}"
                    qw, scales, zeros, g2 = T.rand_case(rng, K, N, bits, gs, orc.F16, 0)
                    zeros = torch.randn(zeros.shape, generator=g2).half() * 0.05
                    q_perm = (torch.randperm(K, generator=g2) if perm else torch.zeros(K)).to(torch.short)
                    Wd = q_linear_cuda.mbwq_q42fp_weight(qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, bits, q_perm.to(DEV))
                    Wo = orc.mbwq_q4_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy() if perm else None, bits, gs)
                    if not np.array_equal(orc.torch_to_np(Wd), Wo):
                        bad.append(tag + ": dequantised weight not bit-exact")
                        continue
                    x = torch.randn((M, K), generator=g2).half()
                    y = q_linear_cuda.mbwq_q4_forward(x.to(DEV), qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, q_perm.to(DEV), bits)
                    T.assert_close(y, T.t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16), orc.F16, tag)
                    good = True
                elif op in ("grad", "pack"):
                    w_bit = int(rng.choice([1, 2, 4, 4, 8]))
                    gs = int(rng.choice([32, 64, 128]))
                    dt = orc.F16 if rng.random() < 0.5 else orc.BF16
                    asym = bool(rng.random() < 0.4)
                    K = gs * int(rng.integers(1, 1024 // gs + 1))
                    N = 32 * int(rng.integers(1, 20)) if asym else int(rng.integers(1, 500))
                    qw, scales, zeros, g2 = T.rand_case(rng, K, N, w_bit, gs, dt, asym)
                    g_idx = torch.arange(K, dtype=torch.int32) // gs
                    if rng.random() < 0.4:
                        g_idx = g_idx[torch.randperm(K, generator=g2)]
                    if op == "grad":
                        M = int(rng.choice([1, 2, 9, 33, 100, 300]))
                        tag = f"grad_input w{w_bit} g{gs} asym={asym} K={K} N={N} M={M}"
                        gy = torch.randn((M, N), generator=g2).to(T.TDT[dt])
                        gx = q_linear_cuda.mpq_grad_input(qw.to(DEV), scales.to(DEV), zeros.to(DEV), g_idx.to(DEV), gy.to(DEV), 16, w_bit, asym)
                        ref = orc.mpq_grad_input(orc.torch_to_np(gy), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros) if not asym else zeros.numpy(),
                                                 g_idx.numpy(), w_bit, gs, int(asym), dt)
                        T.assert_close(gx, T.t16(ref, dt), dt, tag)
                        good = True
                    else:
                        tag = f"dequant/pack w{w_bit} g{gs} asym={asym} K={K} N={N}"
                        Wd = q_linear_cuda.mpq_dequant(qw.to(DEV), scales.to(DEV), zeros.to(DEV), g_idx.to(DEV), w_bit, asym, gs)
                        Wo = orc.mpq_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros) if not asym else zeros.numpy(), g_idx.numpy(), w_bit, gs, int(asym), dt)
                        good = np.array_equal(orc.torch_to_np(Wd), Wo)
                        if good:  # and back: packing the dequantised weight reproduces the packed words where the reference's own round trip does
                            back = q_linear_cuda.mpq_pack(Wd, scales.to(DEV), zeros.to(DEV), g_idx.to(DEV), w_bit, asym, gs)
                            want = orc.mpq_pack(Wo, orc.torch_to_np(scales), orc.torch_to_np(zeros) if not asym else zeros.numpy(), g_idx.numpy(), w_bit, gs, int(asym), dt)
                            good = np.array_equal(back.cpu().numpy(), want)
                else:  # grouped decode call: 2..8 weight sets on one x
                    w_bit = 4 if rng.random() < 0.8 else 2
                    gs = int(rng.choice([32, 64, 128])) if w_bit == 4 else int(rng.choice([64, 128]))
                    dt = orc.F16 if rng.random() < 0.5 else orc.BF16
                    asym = bool(rng.random() < 0.3)
                    K = gs * int(rng.integers(1, 1024 // gs + 1))
                    M = int(rng.integers(1, 17)) if w_bit == 4 else int(rng.integers(1, 3))
                    n = int(rng.integers(2, 9))
                    Ns = [4 * int(rng.integers(1, 120)) for _ in range(n)]
                    if asym:
                        Ns = [max(32 // w_bit, v // (32 // w_bit) * (32 // w_bit)) for v in Ns]
                    tag = f"grouped w{w_bit} g{gs} asym={asym} {'f16' if dt == orc.F16 else 'bf16'} M={M} K={K} N={Ns}"
                    x = torch.randn((M, K), generator=gen).to(T.TDT[dt])
                    sets, hosts = [], []
                    for N in Ns:
                        qw, scales, zeros, _ = T.rand_case(rng, K, N, w_bit, gs, dt, asym)
                        sets.append((qw.to(DEV), scales.to(DEV), zeros.to(DEV), None))
                        hosts.append((qw, scales, zeros))
                    outs = q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, w_bit, asym, gs)
                    for o, (qw, scales, zeros) in zip(outs, hosts):
                        T.assert_close(o, T.oracle_forward(x, qw, scales, zeros, None, w_bit, gs, asym, dt), dt, tag)
                    good = True
                torch.cuda.synchronize()
            except RuntimeError as e:
                key = str(e)[:100]
                refused[key] = refused.get(key, 0) + 1
                continue
            except AssertionError as e:
                bad.append(str(e)[:300])
                continue
            if good:
                ok += 1
            else:
                bad.append(tag + ": not bit-exact")
        res[op] = {"ok": ok, "refused": refused, "bad": bad}
    return {"cases_per_op": cases, "seed": seed, "ops": res}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1), indent=1))
