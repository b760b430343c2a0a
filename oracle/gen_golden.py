"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

* The Python reference (/root/reference) is imported with an in-memory stub of the un-installed
  external `bitorch` package (names only; no behaviour) -- SURVEY.md section 8c.
* The reference's binary CPU extensions are the ones compiled by oracle/Makefile into oracle/_ref/.

The fixtures are DATA: seeded inputs + the reference's outputs.  No reference source is copied.
Run:  python oracle/gen_golden.py      (needs /root/reference; never runs on the GPU box)
"""
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.environ.get("BIE_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))  # BIE_GOLDEN_OUT: regenerate elsewhere and diff
REF = os.environ.get("BIE_REFERENCE", "/root/reference")


def _stub_bitorch():
    """Names the reference imports from the external `bitorch` package (not installed here)."""
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    b = mod("bitorch")
    layers = mod("bitorch.layers")
    ext = mod("bitorch.layers.extensions")
    ql = mod("bitorch.layers.qlinear")
    reg = mod("bitorch.layers.register")
    quant = mod("bitorch.quantizations")
    cfg = mod("bitorch.layers.config")
    mod("bitorch.layers.qconv")

    class RuntimeMode:
        INFERENCE_AUTO = 1
        CPU = 2
        GPU = 4
        DEFAULT = 8

    class _Base(torch.nn.Module):
        pass

    class CustomImplementationMixin:
        pass

    class LayerRecipe:
        pass

    def deco(*a, **k):
        def wrap(cls):
            return cls
        return wrap

    b.RuntimeMode = RuntimeMode
    b.layers = layers
    layers.extensions = ext
    ext.LayerRecipe = LayerRecipe
    ext.CustomImplementationMixin = CustomImplementationMixin
    layers.qlinear = ql
    ql.QLinearBase = _Base
    ql.QLinearImplementation = deco
    layers.register = reg
    reg.QLinearImplementation = deco
    reg.QConv2dImplementation = deco
    layers.QLinearBase = _Base
    layers.CustomImplementationMixin = CustomImplementationMixin
    layers.QConv2dBase = _Base
    layers.QEmbedding = _Base
    layers.QEmbeddingBag = _Base
    layers.config = cfg
    cfg.Config = object
    b.quantizations = quant
    quant.Sign = _Base
    quant.SwishSign = _Base
    quant.Quantization = _Base


def u16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def tonp(t):
    if t is None:
        return None
    if t.dtype in (torch.float16, torch.bfloat16):
        return u16(t)
    return t.detach().cpu().contiguous().numpy()


def update_step_vectors():
    from bitorch_engine.layers.qlinear.nbit.layer import MPQWeightParameter
    from bitorch_engine.utils.model_helper import qweight_update_fn
    out = {}
    K, N, w_bit, gs = 128, 64, 4, 32
    for tag, dtype in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for order in ("trivial", "actorder"):
            g = torch.Generator().manual_seed(31 + len(tag) + len(order))
            qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
            scales = (torch.rand((K // gs, N), generator=g) * 0.01 + 0.005).to(dtype)
            qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // gs, N * w_bit // 32), generator=g, dtype=torch.int64).to(torch.int32)
            g_idx = torch.arange(K, dtype=torch.int32) // gs
            if order == "actorder":
                g_idx = g_idx[torch.randperm(K, generator=g)]
            p = MPQWeightParameter(qweight.clone(), requires_grad=False, scales=scales.clone(), zeros=qzeros.clone(), g_idx=g_idx.clone(),
                                   w_bit=w_bit, asym=True, group_size=gs, layer_type=1)
            exp_l, exp_s = torch.zeros((K, N), dtype=dtype), torch.zeros((K, N), dtype=dtype)
            step = torch.tensor(0.0)
            key = f"{tag}_{order}"
            out[key + "_qweight0"], out[key + "_scales"], out[key + "_qzeros0"], out[key + "_g_idx"] = tonp(qweight), tonp(scales), tonp(qzeros), tonp(g_idx)
            for it in range(1, 6):
                grad = (torch.randn((K, N), generator=g) * 0.02).to(dtype)
                out[f"{key}_grad{it}"] = tonp(grad)
                qweight_update_fn(p, exp_avg_s=exp_s, exp_avg_l=exp_l, step=step, lr=2e-3, weight_decay=0.0, beta1=0.9, beta2=0.99, eps=1e-6,
                                  dtype=dtype, correct_bias=(it % 2 == 0), projector=None, grad=grad)
                out[f"{key}_qweight{it}"] = tonp(p.data)
                out[f"{key}_exp_l{it}"], out[f"{key}_exp_s{it}"] = tonp(exp_l).copy(), tonp(exp_s).copy()  # snapshots: the moments are updated in place
            out[key + "_qzeros5"] = tonp(p.zeros)
            out[key + "_meta"] = np.array([K, N, w_bit, gs])
    return out


def update_step_integer_params_vectors():
    """The other branches of the reference's qweight_update_fn (utils/model_helper.py:403-478): binary linear / conv parameters (sign flips
    from two lerp-ed moments), W4A4 / W8A8 integer parameters (Adam on the integer values, re-quantised with nv_tensor_quant) and the
    boolean binary embedding table (XOR with the moment's sign bits on the active rows).  The integer gradients are ASSIGNED to `.grad`
    (stock torch cannot produce them; the reference runs a patched torch), three steps each, fp16 and bf16 moments."""
    from bitorch_engine.layers.qlinear.binary import BinaryLinearParameter
    from bitorch_engine.layers.qconv.binary import BinaryConvParameter
    from bitorch_engine.layers.qlinear.nbit import nBitLinearParameter
    from bitorch_engine.layers.qconv.nbit import nBitConvParameter
    from bitorch_engine.layers.qembedding.binary import BinaryEmbeddingParameter
    from bitorch_engine.utils.model_helper import qweight_update_fn
    out = {}
    plain = lambda cls, data: torch.Tensor._make_subclass(cls, data, False)  # stock torch: no requires_grad on integer data
    for tag, dtype in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for kind, cls, shape in (("binlin", BinaryLinearParameter, (24, 40)), ("binconv", BinaryConvParameter, (6, 8, 3, 3)),
                                 ("nbitlin", nBitLinearParameter, (24, 40)), ("nbitconv", nBitConvParameter, (6, 8, 3, 3)),
                                 ("binemb", BinaryEmbeddingParameter, (16, 32))):
            g = torch.Generator().manual_seed(77 + len(kind) * 5 + len(tag))
            key = f"{tag}_{kind}"
            if kind.startswith("bin") and kind != "binemb":
                data = torch.where(torch.rand(shape, generator=g) > 0.5, 1, -1).to(torch.int8)
            elif kind == "binemb":
                data = torch.rand(shape, generator=g) > 0.5
            else:
                data = torch.randint(-7, 8, shape, generator=g).to(torch.int8)
            p = plain(cls, data.clone())
            if kind == "binemb":
                p.active_indices = torch.tensor([1, 3, 4, 9, 15])
                out[key + "_active"] = tonp(p.active_indices)
            out[key + "_w0"] = tonp(data)
            exp_l, exp_s = torch.zeros(shape, dtype=dtype), torch.zeros(shape, dtype=dtype)
            step = torch.tensor(0.0)
            for it in range(1, 4):
                if kind == "binemb":
                    grad = torch.rand(shape, generator=g) > 0.5
                else:
                    grad = torch.randint(-5, 6, shape, generator=g).to(torch.int8)
                p.grad = None
                p.grad_dtype = None  # the W4A4 / W8A8 branch re-types the data (nv_tensor_quant returns `dtype`); the gradient stays integer
                p.grad = grad
                out[f"{key}_grad{it}"] = tonp(grad)
                qweight_update_fn(p, exp_avg_s=exp_s, exp_avg_l=exp_l, step=step, lr=3e-2, weight_decay=(0.01 if it == 2 else 0.0), beta1=0.9, beta2=0.99,
                                  eps=1e-6, dtype=dtype, correct_bias=(it % 2 == 1), projector=None, grad=None)
                out[f"{key}_w{it}"] = tonp(p.data).copy()  # a snapshot: the binary branches update the data in place
                out[f"{key}_wdtype{it}"] = np.array([str(p.data.dtype)])
                out[f"{key}_exp_l{it}"], out[f"{key}_exp_s{it}"] = tonp(exp_l).copy(), tonp(exp_s).copy()
            out[key + "_step"] = np.array([float(step)])
    return out


def optim_diodemix_vectors():
    """SURVEY 8f-2, the caller of the update step: the reference's DiodeMix optimiser (optim/diode_beta.py:37-196) and GaLoreProjector
    (optim/galore_projector.py:17-124) run on CPU tensors.  Float parameters (the AdamW branch, two groups), GaLore groups of every projection
    type, a binary linear and a W4A4-style integer parameter through the optimiser (the sign carriers' first moments are drawn with
    torch.rand_like under torch.manual_seed: a mirror must consume the generator identically), and an MPQ parameter fed by privileged_grad."""
    from bitorch_engine.optim import DiodeMix
    from bitorch_engine.layers.qlinear.binary import BinaryLinearParameter
    from bitorch_engine.layers.qlinear.nbit import nBitLinearParameter, MPQWeightParameter
    out = {}
    plain = lambda cls, data: torch.Tensor._make_subclass(cls, data, False)
    # ---- float parameters: two groups with different options, four steps
    g = torch.Generator().manual_seed(501)
    p1 = torch.nn.Parameter(torch.randn((24, 40), generator=g))
    p2 = torch.nn.Parameter(torch.randn((40,), generator=g))
    p3 = torch.nn.Parameter(torch.randn((8, 8), generator=g))
    out["float_p1_0"], out["float_p2_0"], out["float_p3_0"] = tonp(p1).copy(), tonp(p2).copy(), tonp(p3).copy()  # snapshots: step() updates in place
    opt = DiodeMix([{"params": [p1, p2], "weight_decay": 0.01}, {"params": [p3], "lr": 5e-4, "correct_bias": False}], lr=1e-3, betas=(0.9, 0.99), eps=1e-6)
    torch.manual_seed(9001)
    for it in range(1, 5):
        for name, p in (("p1", p1), ("p2", p2), ("p3", p3)):
            p.grad = torch.randn(p.shape, generator=g) * 0.1
            out[f"float_{name}_grad{it}"] = tonp(p.grad)
        opt.step()
        for name, p in (("p1", p1), ("p2", p2), ("p3", p3)):
            out[f"float_{name}_{it}"] = tonp(p).copy()
    out["float_p1_m"], out["float_p1_v"] = tonp(opt.state[p1]["exp_avg_l"]), tonp(opt.state[p1]["exp_avg_s"])
    # ---- GaLore groups: every projection type, tall and wide gradients, the projector refreshed every second step
    for tag, pt, shape in (("std_tall", "std", (48, 20)), ("std_wide", "std", (20, 48)), ("rstd_tall", "reverse_std", (48, 20)), ("rstd_wide", "reverse_std", (20, 48)),
                           ("left", "left", (20, 48)), ("right", "right", (20, 48)), ("full", "full", (20, 48))):
        g = torch.Generator().manual_seed(600 + len(tag))
        p = torch.nn.Parameter(torch.randn(shape, generator=g))
        out[f"galore_{tag}_0"] = tonp(p).copy()
        opt = DiodeMix([{"params": [p], "rank": 4, "update_proj_gap": 2, "scale": 0.25, "proj_type": pt}], lr=2e-3, betas=(0.9, 0.99), weight_decay=0.0)
        for it in range(1, 5):
            p.grad = torch.randn(shape, generator=g) * 0.1
            out[f"galore_{tag}_grad{it}"] = tonp(p.grad)
            opt.step()
            out[f"galore_{tag}_{it}"] = tonp(p).copy()
    # ---- quantised parameters through the optimiser: binary linear (sign flips) and W4A4-style integer values, moments in fp32 and bf16
    for tag, dtype in (("f32", torch.float), ("bf16", torch.bfloat16)):
        for kind, cls in (("binlin", BinaryLinearParameter), ("nbitlin", nBitLinearParameter)):
            g = torch.Generator().manual_seed(700 + len(kind) + len(tag))
            shape = (24, 40)
            data = torch.where(torch.rand(shape, generator=g) > 0.5, 1, -1).to(torch.int8) if kind == "binlin" else torch.randint(-7, 8, shape, generator=g).to(torch.int8)
            p = plain(cls, data.clone())
            key = f"q_{tag}_{kind}"
            out[key + "_w0"] = tonp(data)
            opt = DiodeMix([p], lr=3e-2, betas=(0.9, 0.99), eps=1e-6, weight_decay=0.0, dtype=dtype)
            torch.manual_seed(4242)  # the first step draws the sign carriers' starting moments
            for it in range(1, 4):
                grad = torch.randint(-5, 6, shape, generator=g).to(torch.int8)
                p.grad = None
                p.grad_dtype = None
                p.grad = grad
                out[f"{key}_grad{it}"] = tonp(grad)
                opt.step()
                out[f"{key}_w{it}"] = tonp(p.data).copy()
                out[f"{key}_wdtype{it}"] = np.array([str(p.data.dtype)])
                out[f"{key}_exp_s{it}"] = tonp(opt.state[p]["exp_avg_s"]).copy()
                out[f"{key}_exp_l{it}"] = tonp(opt.state[p]["exp_avg_l"]).copy()
    # ---- an MPQ parameter (GPTQ form) fed through privileged_grad, five steps (the fifth also moves the zero points)
    K, N, w_bit, gs = 128, 64, 4, 32
    g = torch.Generator().manual_seed(811)
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    scales = (torch.rand((K // gs, N), generator=g) * 0.01 + 0.005).to(torch.float16)
    qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // gs, N * w_bit // 32), generator=g, dtype=torch.int64).to(torch.int32)
    g_idx = torch.arange(K, dtype=torch.int32) // gs
    p = MPQWeightParameter(qweight.clone(), requires_grad=False, scales=scales.clone(), zeros=qzeros.clone(), g_idx=g_idx.clone(), w_bit=w_bit, asym=True,
                           group_size=gs, layer_type=1)
    out["mpq_qweight0"], out["mpq_scales"], out["mpq_qzeros0"], out["mpq_g_idx"] = tonp(qweight), tonp(scales), tonp(qzeros), tonp(g_idx)
    out["mpq_meta"] = np.array([K, N, w_bit, gs])
    opt = DiodeMix([p], lr=2e-3, betas=(0.9, 0.99), eps=1e-6, dtype=torch.float16)
    for it in range(1, 6):
        p.privileged_grad = (torch.randn((K, N), generator=g) * 0.02).to(torch.float16)
        p.grad = torch.zeros_like(p.data)  # step() skips parameters without .grad; the MPQ branch reads privileged_grad
        out[f"mpq_grad{it}"] = tonp(p.privileged_grad)
        opt.step()
        out[f"mpq_qweight{it}"] = tonp(p.data).copy()
    out["mpq_qzeros5"] = tonp(p.zeros)
    return out



def extension_signatures():
    import glob
    import re
    mods = {"q_linear_cuda": "layers/qlinear/nbit/cuda/q_linear_cuda.cpp", "q_linear_cutlass": "layers/qlinear/nbit/cutlass/q_linear_cutlass.cpp",
            "binary_linear_cpp": "layers/qlinear/binary/cpp/binary_linear.cpp", "binary_linear_cuda": "layers/qlinear/binary/cuda/binary_linear_cuda.cpp",
            "binary_linear_cutlass": "layers/qlinear/binary/cutlass/binary_linear_cutlass.cpp", "binary_conv_cpp": "layers/qconv/binary/cpp/binary_conv.cpp",
            "binary_conv2d_cutlass": "layers/qconv/binary/cutlass/binary_conv2d_cutlass.cpp", "q4_conv_cutlass": "layers/qconv/nbit/cutlass/q4_conv_cutlass.cpp",
            "functions_cuda": "functions/cuda/functions_cuda.cpp"}
    out = {}
    for mod, rel in mods.items():
        text = open(os.path.join(REF, "bitorch_engine", rel)).read()
        text_nc = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text_nc = re.sub(r"//[^\n]*", "", text_nc)
        fns = {}
        for name, fn in re.findall(r'm\.def\(\s*"(\w+)"\s*,\s*&(\w+)', text_nc):
            m = None
            for m in re.finditer(r"\b" + fn + r"\s*\(([^)]*)\)\s*\{", text_nc):
                pass  # the last match with a body is the definition
            if m is None:  # defined in the .cu: the .cpp holds the declaration
                m = re.search(r"\b" + fn + r"\s*\(([^)]*)\)\s*;", text_nc)
            assert m is not None, (mod, fn)
            params = [re.sub(r"\s*=.*$", "", a).strip().split()[-1].lstrip("&*") for a in m.group(1).split(",") if a.strip()]
            fns[name] = params
        out[mod] = fns
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    _stub_bitorch()
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    if len(sys.argv) > 1 and sys.argv[1] == "optim":  # this one fixture only
        np.savez_compressed(os.path.join(OUT, "optim_diodemix.npz"), **optim_diodemix_vectors())
        print("written", os.path.join(OUT, "optim_diodemix.npz"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "update_int":  # this one fixture only (the others stay byte-identical)
        np.savez_compressed(os.path.join(OUT, "update_step_integer_params.npz"), **update_step_integer_params_vectors())
        print("written", os.path.join(OUT, "update_step_integer_params.npz"))
        return
    from bitorch_engine.layers.qlinear.nbit import MPQWeightParameter
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import unpack_qweight, pack_fp_weight, make_group_map
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda, MBWQLinearCuda
    from bitorch_engine.utils.quant_operators import gptq_style_zeros_packing

    g = torch.Generator().manual_seed(1234)
    manifest = {}

    # ------------------------------------------------------------------ 1. unpack_qweight / pack_fp_weight
    cases = []
    for dt_name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for w_bit in (1, 2, 4, 8):
            for mode in ("sym_gidx", "sym_nogidx", "sym_actorder", "asym"):
                if mode == "sym_nogidx" and w_bit != 4:
                    continue
                if mode == "sym_actorder" and w_bit not in (2, 4):
                    continue
                for (K, N, gs) in ((128, 32, 64), (64, 64, 32)):
                    G = K // gs
                    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
                    scales = (torch.rand((G, N), generator=g) * 0.01 + 0.005).to(dt)
                    maxq = 2 ** w_bit - 1
                    g_idx = torch.tensor([i // gs for i in range(K)], dtype=torch.int32)
                    if mode == "sym_actorder":
                        g_idx = g_idx[torch.randperm(K, generator=g)]
                    p = MPQWeightParameter(qweight.clone(), requires_grad=False, w_bit=w_bit, asym=(mode == "asym"),
                                           group_size=gs, layer_type=1)
                    p.scales = scales
                    p.g_idx = None if mode == "sym_nogidx" else g_idx
                    if mode == "asym":
                        if (N * w_bit) % 32:
                            continue
                        p.zeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (G, N * w_bit // 32), generator=g, dtype=torch.int64).to(torch.int32)
                    else:
                        p.zeros = (scales.float() * torch.rand((G, N), generator=g) * maxq).to(dt)
                    W = unpack_qweight(p)
                    assert W.dtype == dt, (W.dtype, dt)
                    # pack_fp_weight of a *perturbed* dense weight (exercises rounding + clamping)
                    Wp = (W.float() + (torch.rand(W.shape, generator=g) - 0.5) * 0.004).to(dt)
                    packed = pack_fp_weight(Wp, p)
                    name = f"mpq_{dt_name}_w{w_bit}_{mode}_K{K}N{N}g{gs}"
                    cases.append(name)
                    np.savez_compressed(os.path.join(OUT, name + ".npz"), qweight=tonp(qweight), scales=tonp(scales),
                                        zeros=tonp(p.zeros), g_idx=(np.zeros(0, np.int32) if p.g_idx is None else tonp(p.g_idx)),
                                        W=tonp(W), Wp=tonp(Wp), packed=tonp(packed),
                                        meta=np.array([K, N, gs, w_bit, int(mode == "asym"), int(p.g_idx is not None)], np.int64))
    manifest["mpq_dequant_pack"] = cases

    # gptq_style_zeros_packing
    zq = torch.randint(0, 16, (4, 64), generator=g)
    zp = gptq_style_zeros_packing(zq.clone(), 4, 64, 64)
    np.savez_compressed(os.path.join(OUT, "gptq_zeros_packing.npz"), zq=tonp(zq), packed=tonp(zp))

    # ------------------------------------------------------------------ 2. full MPQLinearCuda layers (CPU path, M > 32)
    layer_cases = []
    sd_tables = {}
    for name, kw in (
        ("gba_sym_w4_g128_dq2", dict(w_bit=4, dtype=torch.half, group_size=128, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False)),
        ("gba_sym_w2_g32_dq1", dict(w_bit=2, dtype=torch.half, group_size=32, dq_group_size=1, dq_mode=1, use_gba_quant=True, asym=False)),
        ("gba_sym_w4_g128_bf16", dict(w_bit=4, dtype=torch.bfloat16, group_size=128, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False)),
        ("gba_asym_w4_g64", dict(w_bit=4, dtype=torch.half, group_size=64, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=True)),
        ("gptq_w4_g64", dict(w_bit=4, dtype=torch.half, group_size=64, use_gba_quant=False, asym=True)),
        ("gptq_w8_g128", dict(w_bit=8, dtype=torch.half, group_size=128, use_gba_quant=False, asym=True)),
        ("gba_sym_w4_g256_nodq", dict(w_bit=4, dtype=torch.half, group_size=256, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False)),
    ):
        K, N = 256, 128
        layer = MPQLinearCuda(in_channels=K, out_channels=N, requires_grad=False, **kw)
        sd_tables["MPQLinearCuda/" + name] = {k: [list(v.shape), str(v.dtype)] for k, v in layer.state_dict().items()}
        sd = {}
        for k, v in layer.state_dict().items():
            if k == "qweight" or (k == "qzeros" and v.dtype == torch.int32):
                nv = torch.randint(-2 ** 31, 2 ** 31 - 1, v.shape, generator=g, dtype=torch.int64).to(torch.int32)
            elif v.dtype == torch.uint8:
                nv = torch.randint(0, 256, v.shape, generator=g, dtype=torch.int64).to(torch.uint8)
            elif k in ("g_idx", "wf"):
                nv = v.clone()
            elif k == "bias":
                nv = torch.zeros_like(v)
            elif "scales" in k:
                nv = (torch.rand(v.shape, generator=g) * 0.01 + 0.002).to(v.dtype)
            else:  # *_zeros, zeros
                nv = (torch.rand(v.shape, generator=g) * 4).to(v.dtype)
            sd[k] = nv
        layer.load_state_dict(sd)
        layer.qweight.data = sd["qweight"]
        layer.prepare_params()
        outs = {}
        for M in (33, 64):
            x = torch.randn((M, K), generator=g).to(kw["dtype"])
            y = layer(x)
            outs[f"x{M}"] = tonp(x)
            outs[f"y{M}"] = tonp(y)
        np.savez_compressed(os.path.join(OUT, f"layer_{name}.npz"),
                            **{"sd_" + k: tonp(v) for k, v in sd.items()},
                            prep_scales=tonp(layer.scales), prep_zeros=tonp(layer.zeros), **outs)
        layer_cases.append(name)
    manifest["mpq_layers"] = layer_cases

    # MBWQ constructor state_dict tables + make_group_map
    for name, kw in (
        ("q4", dict(w_bit=4, dtype=torch.half, group_size=32, dq_group_size=1, use_gba_quant=True, asym=False, dq_mode=2, use_mbw=False)),
        ("exl2", dict(w_bit=4, dtype=torch.half, group_size=32, dq_group_size=1, use_gba_quant=True, asym=False, dq_mode=2, use_mbw=True, groups=8, rows_packed=24)),
    ):
        layer = MBWQLinearCuda(in_channels=256, out_channels=128, requires_grad=False, **kw)
        sd_tables["MBWQLinearCuda/" + name] = {k: [list(v.shape), str(v.dtype)] for k, v in layer.state_dict().items()}

    spec = importlib.util.spec_from_file_location("ref_test_util", os.path.join(REF, "tests", "layers", "util.py"))
    ref_util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_util)
    gm = {}
    for cname, K, bits, prop, gsz in (
        ("q_proj", 256, [4, 2], [0.75, 0.25], {"4": 32, "2": 32}),
        ("k_proj", 256, [4, 2], [0.25, 0.75], {"4": 32, "2": 32}),
        ("w3w2", 512, [3, 2], [0.5, 0.5], {"3": 32, "2": 32}),
        ("all6", 1024, [8, 6, 5, 4, 3, 2], [0.125, 0.125, 0.125, 0.25, 0.125, 0.25], {"8": 32, "6": 32, "5": 32, "4": 32, "3": 32, "2": 32}),
    ):
        groups, rows = ref_util.get_packed_info(K, bits, prop, gsz)
        qg = ref_util.get_q_groups(groups, bits, gsz, K, prop)
        gmap = make_group_map(torch.tensor(qg, dtype=torch.short), rows)
        gm[cname + "_meta"] = np.array([K, groups, rows], np.int64)
        gm[cname + "_q_groups"] = np.array(qg, np.int16)
        gm[cname + "_group_map"] = tonp(gmap)
    np.savez_compressed(os.path.join(OUT, "exl2_group_maps.npz"), **gm)

    with open(os.path.join(OUT, "state_dict_tables.json"), "w") as f:
        json.dump(sd_tables, f, indent=1, sort_keys=True)

    # ------------------------------------------------------------------ 3. binary CPU extensions (compiled reference)
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import binary_linear_cpp
    import binary_conv_cpp
    bl = {}
    for (M, N, K) in ((1, 64, 128), (4, 96, 256), (33, 40, 64)):
        x = torch.randn((M, K), generator=g)
        x[0, :4] = 0.0  # sign(0) must count as +1
        w = torch.randn((N, K), generator=g)
        y = binary_linear_cpp.forward(x, w, M, N, K)
        wp = binary_linear_cpp.w_pack(w, N, K)
        y2 = binary_linear_cpp.forward(x, wp, M, N, K)
        assert torch.equal(y, y2)
        tag = f"M{M}N{N}K{K}"
        bl[tag + "_x"], bl[tag + "_w"], bl[tag + "_y"], bl[tag + "_wpacked"] = tonp(x), tonp(w), tonp(y), tonp(wp)
    np.savez_compressed(os.path.join(OUT, "binary_linear_cpp.npz"), **bl)

    bc = {}
    for (B, C, H, OC, ks, st, pad, dil) in ((2, 16, 7, 8, 3, 1, 1, 1), (1, 8, 9, 16, 3, 2, 1, 1), (2, 8, 8, 8, 1, 1, 0, 1), (1, 8, 10, 8, 3, 1, 2, 2)):
        x = torch.randn((B, C, H, H), generator=g)
        w = torch.randn((OC, C, ks, ks), generator=g)
        oe = (H + 2 * pad - dil * (ks - 1) - 1) // st + 1
        k_ = C * ks * ks
        y = binary_conv_cpp.forward(x, w.view(OC, -1).contiguous(), OC, oe * oe, k_, ks, st, pad, dil, oe)
        tag = f"B{B}C{C}H{H}OC{OC}k{ks}s{st}p{pad}d{dil}"
        bc[tag + "_x"], bc[tag + "_w"], bc[tag + "_y"] = tonp(x), tonp(w), tonp(y)
    np.savez_compressed(os.path.join(OUT, "binary_conv_cpp.npz"), **bc)

    # ------------------------------------------------------------------ 4. known-answer vector held by the reference's tests
    # tests/functions/test_quant_ops.py:110-157 : bytes [0,16,35,255] -> +-1, LSB first
    kat_bytes = np.array([0, 16, 35, 255], np.uint8)
    kat_expected = np.array([-1] * 8 + [-1, -1, -1, +1, -1, -1, -1, -1][::-1] + [-1, -1, +1, -1, -1, -1, +1, +1][::-1] + [1] * 8, np.float32)
    np.savez_compressed(os.path.join(OUT, "kat_unpack_uint8.npz"), bytes=kat_bytes, expected=kat_expected)

    # ------------------------------------------------------------------ 5. W4A4 / W8A8 quantisers (python reference)
    # utils/quant_operators.py:234-305.  The CUDA kernels round half AWAY from zero (roundf) where torch.round is
    # half-to-even, so exact .5 ties are excluded from the vectors (flagged in `tie`).
    from bitorch_engine.utils.quant_operators import q4_quantization, q8_quantization
    qq = {}
    x = torch.randn((24, 64), generator=g) * 1.7
    eps = torch.tensor(0.00001)
    q4d, s4 = q4_quantization(x, None, eps)
    q8d, s8 = q8_quantization(x, None, eps)
    sa = torch.tensor(0.37)
    qq["x"], qq["q4_derived"], qq["scale4"], qq["q8_derived"], qq["scale8"] = tonp(x), tonp(q4d), tonp(s4), tonp(q8d), tonp(s8)
    qq["scale_given"] = tonp(sa)
    qq["q4_given"], qq["q8_given"] = tonp(q4_quantization(x, sa, eps)), tonp(q8_quantization(x, sa, eps))
    r = x / s4
    qq["tie"] = tonp((r - r.floor()) == 0.5)
    np.savez_compressed(os.path.join(OUT, "q4_q8_quantization.npz"), **qq)

    # ------------------------------------------------------------------ 6. helpers either side of the path (python reference)
    # utils/quant_operators.py:7-90 (nv_tensor_quant), utils/model_helper.py:54-155,286-327, utils/convert.py:94-119,
    # layers/qembedding/binary/layer.py:343-556 (BinaryEmbeddingBag: pure torch, CPU-runnable).  A separate generator keeps the
    # earlier sections' random streams untouched.
    from bitorch_engine.utils.quant_operators import nv_tensor_quant
    from bitorch_engine.utils import model_helper as mh
    from bitorch_engine.utils.convert import get_mpq_config
    g6 = torch.Generator().manual_seed(606)
    hp = {}
    xq = torch.randn((6, 16), generator=g6) * 2.5
    for tag, kw in (("default", {}), ("bits4", {"num_bits": 4}), ("wide", {"narrow_range": False}),
                    ("amax_rows", {"amax": xq.abs().amax(dim=1, keepdim=True)})):
        q, sc = nv_tensor_quant(xq.clone(), **kw)
        hp[f"nvq_{tag}_q"], hp[f"nvq_{tag}_scale"] = tonp(q), tonp(sc)
    qh, sh = nv_tensor_quant(xq.to(torch.bfloat16))
    hp["nvq_x"], hp["nvq_bf16_q"], hp["nvq_bf16_scale"] = tonp(xq), u16(qh), tonp(sh)
    qu, su = nv_tensor_quant(xq.abs(), unsigned=True)
    hp["nvq_unsigned_q"], hp["nvq_unsigned_scale"] = tonp(qu), tonp(su)
    wi = torch.randn((8, 16), generator=g6)
    wq, ws = mh.init_weight(wi.clone(), cls=lambda t_: torch.nn.Parameter(t_, requires_grad=False))  # stock torch: no grad on int8
    hp["iw_w"], hp["iw_q"], hp["iw_scale"] = tonp(wi), tonp(wq.data), tonp(ws)
    t = torch.randn((2, 5, 130), generator=g6)
    tp, added = mh.pad_last_2_dims_to_multiple_of_128(t)
    hp["pad_in"], hp["pad_out"], hp["pad_added"] = tonp(t), tonp(tp), np.array([added])
    pop = torch.randint(0, 64, (6, 8, 12), generator=g6)
    hp["post_in"] = tonp(pop)
    hp["post_out"] = tonp(mh.binary_matmul_forward_post_processing(pop.clone(), [2, 3], 3, 4, 64))
    we = torch.randn((5, 13), generator=g6)
    hp["emb_in"], hp["emb_padded"] = tonp(we), tonp(mh.pad_embedding_dim(we))
    from bitorch_engine.layers.qembedding.binary.layer import BinaryEmbeddingBagForward
    table = torch.rand((20, 24), generator=g6) > 0.5
    idx = torch.randint(0, 20, (4, 5), generator=g6)
    hp["bag_table"], hp["bag_idx"] = tonp(table), tonp(idx)
    hp["bag_out"] = tonp(BinaryEmbeddingBagForward.apply(idx, table, False))
    np.savez_compressed(os.path.join(OUT, "helpers.npz"), **hp)
    manifest["mpq_configs"] = {k: get_mpq_config(k) for k in (None, "2-8-32", "2-32-32", "2-128-32", "4-128-256", "8-128-256")}
    manifest["mpq_configs"] = {str(k): v for k, v in manifest["mpq_configs"].items()}

    # ---- the DiodeMix re-pack step, MPQWeightParameter.update -> qweight_update_fn (utils/model_helper.py:363-532) on a
    # GPTQ-style (asym, g_idx) parameter: five steps each (step 5 also runs update_zeros), trivial and permuted g_idx, fp16 and bf16
    np.savez_compressed(os.path.join(OUT, "update_step.npz"), **update_step_vectors())
    np.savez_compressed(os.path.join(OUT, "update_step_integer_params.npz"), **update_step_integer_params_vectors())
    np.savez_compressed(os.path.join(OUT, "optim_diodemix.npz"), **optim_diodemix_vectors())

    # ---- the extension modules' boundary: name and positional parameter list of every function the reference binds with pybind11
    # (m.def("name", &fn)), read off the reference's own C++ definitions.  Data only (names), consumed by tests/test_boundary_cpu.py.
    with open(os.path.join(OUT, "extension_signatures.json"), "w") as f:
        json.dump(extension_signatures(), f, indent=1, sort_keys=True)

    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total bytes:", tot)


if __name__ == "__main__":
    main()
