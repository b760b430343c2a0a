"""Drop-in for the reference's `q_linear_cuda` pybind module
(layers/qlinear/nbit/cuda/q_linear_cuda.cpp:357-369): same function names and argument order."""
import os

import torch

from bitorch_engine import _hip

def _group_size(K, scales):
    G = scales.shape[0]
    return (K + G - 1) // G


def _cached(t, key, compute, also=()):
    """Memoise a property of tensor `t` ON the tensor object, invalidated by in-place modification (`_version`)
    or re-pointing (`data_ptr`) of `t` or of the tensors in `also`.  (A process-wide dict keyed by address would alias
    recycled allocations.)"""
    tag = (t._version, t.data_ptr()) + tuple(v for o in also for v in (o._version, o.data_ptr()))
    memo = getattr(t, "_bie_memo", None)
    if memo is None:
        memo = {}
        try:
            t._bie_memo = memo
        except Exception:
            pass
    hit = memo.get(key)
    if hit is None or hit[0] != tag:
        hit = memo[key] = (tag, compute())
    return hit[1]


def gidx_is_trivial(g_idx, group_size):
    """True when g_idx[k] == k // group_size (what MPQLinearBase initialises, nbit/layer.py:385-386).
    One device->host sync per distinct tensor version, then remembered on the tensor."""
    if g_idx is None:
        return True

    def compute():
        ref = torch.arange(g_idx.numel(), device=g_idx.device, dtype=torch.int32) // group_size
        return bool(torch.equal(g_idx.to(torch.int32), ref))
    return _cached(g_idx, ("trivial", group_size), compute)


def act_order_sorted(qweight, g_idx, w_bit, group_size):
    """Act-order checkpoints whose g_idx is a permutation of k // group_size (what GPTQ's desc_act writes): the packed matrix
    re-ordered once so that every group's k are consecutive (bie_mpq_sort_rows) + the column order to apply to x.  Returns
    (perm int32[K], qweight_sorted) or None (g_idx not of that form, or BIE_ACT_ORDER_SORTED=0).  The sorted copy costs
    K*N*w/8 bytes beside the checkpoint tensor and is remembered per (g_idx, qweight) version."""
    import os
    if os.environ.get("BIE_ACT_ORDER_SORTED", "1") == "0":
        return None

    def compute():
        K = g_idx.numel()
        g = g_idx.to(torch.int64)
        perm = torch.argsort(g, stable=True)
        ref = torch.arange(K, device=g.device, dtype=torch.int64) // group_size
        if not bool(torch.equal(g[perm], ref)):
            return None
        perm = perm.to(torch.int32).contiguous()
        qw = qweight.contiguous()
        out = torch.empty_like(qw)
        rc = _hip.lib().bie_mpq_sort_rows(_hip.ptr(qw), _hip.ptr(perm), _hip.ptr(out), K, qw.shape[1], w_bit, _hip.stream())
        _hip.check(rc, "bie_mpq_sort_rows")
        return perm, out
    return _cached(g_idx, ("sorted", w_bit, group_size), compute, also=(qweight,))


def gather_cols(x, perm):
    """x[:, perm] (bie_gather_cols)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    rc = _hip.lib().bie_gather_cols(_hip.ptr(x), _hip.ptr(perm), _hip.ptr(out), x.shape[0], x.shape[1], _hip.dt(x), _hip.stream())
    _hip.check(rc, "bie_gather_cols")
    return out


def perm_or_none(q_perm):
    """The reference treats an all-zero q_perm as 'no permutation' with a per-call .item() sync
    (mbwq_linear_cuda_kernel.cu:671,777); here the answer is remembered per tensor version."""
    if q_perm is None:
        return None
    zero = _cached(q_perm, "allzero", lambda: bool(torch.all(q_perm == 0).item()))
    return None if zero else q_perm


def mpq_forward_impl(x, qweight, scales, zeros, g_idx, w_bit, asym, group_size, bias=None, trivial_gidx=None, out=None):
    """out: optional contiguous [M, N] tensor of x's dtype that receives y (the kernels write it in place: no copy)."""
    _hip.need_gpu(x, qweight, scales, zeros)
    x = x.contiguous()
    M, K = x.shape
    N = qweight.shape[1]
    if trivial_gidx is None:
        trivial_gidx = gidx_is_trivial(g_idx, group_size)
    if not trivial_gidx and M > 0:
        # act-order checkpoints (explicit g_idx): re-ordered once into an implicit-group matrix, then the fast kernels on x[:, perm]
        so = act_order_sorted(qweight, g_idx, w_bit, group_size)
        if so is not None:
            x, qweight, trivial_gidx = gather_cols(x, so[0]), so[1], True
    # g_idx that is NOT a permutation of k // group_size (unequal groups): decode rows run the generic kernel, prefill (M > 32) the
    # per-k dequantise into the MFMA fragment image + the dense kernel (bie_mpq_forward picks it when the workspace has room for the
    # image: bie_mpq_workspace_bytes_gidx) -- the split the reference makes for every M > 32 (layers/qlinear/nbit/cuda/mpq_layer.py:59-62),
    # without a vendor GEMM
    gptr = None if trivial_gidx else g_idx.to(torch.int32).contiguous()
    pitched = out is not None and out.dim() == 2 and not out.is_contiguous() and out.stride(1) == 1 and out.stride(0) >= N
    if out is not None and (out.shape != (M, N) or out.dtype != x.dtype or out.device != x.device or not (out.is_contiguous() or pitched)):
        raise RuntimeError("mpq_forward_impl: out must be an [M, N] tensor of x's dtype on x's device, contiguous or a column range of a wider "
                           "row-major tensor (unit column stride)")
    y = torch.empty((M, N), dtype=x.dtype, device=x.device) if out is None else out
    if M == 0:
        return y
    L = _hip.lib()
    need = (L.bie_mpq_workspace_bytes if gptr is None else L.bie_mpq_workspace_bytes_gidx)(M, K, N, w_bit)
    ws = _hip.workspace(need, x.device)
    if pitched:
        # the GEMM epilogue stores straight into the column range (bie_mpq_forward_pitched); shapes it does not take: tight buffer + one copy
        rc = -2 if gptr is not None else L.bie_mpq_forward_pitched(
            _hip.ptr(x), _hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()), _hip.ptr(bias), out.data_ptr(), out.stride(0),
            _hip.ptr(ws), 0 if ws is None else ws.numel(), M, K, N, w_bit, group_size, int(bool(asym)), _hip.dt(x), _hip.stream())
        if rc == 0:
            return out
        if rc != -2:  # BIE_ERR_UNSUPPORTED
            _hip.check(rc, "bie_mpq_forward_pitched")
        return out.copy_(mpq_forward_impl(x, qweight, scales, zeros, g_idx, w_bit, asym, group_size, bias, trivial_gidx, None))
    rc = L.bie_mpq_forward(_hip.ptr(x), _hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                           _hip.ptr(gptr), _hip.ptr(bias), _hip.ptr(y), _hip.ptr(ws), 0 if ws is None else ws.numel(),
                           M, K, N, w_bit, group_size, int(bool(asym)), _hip.dt(x), _hip.stream())
    _hip.check(rc, "bie_mpq_forward")
    return y


def mpq_forward_grouped_impl(x, sets, w_bit, asym, group_size):
    """Several layers that share x (q/k/v, gate/up) in ONE decode launch (bie_mpq_forward_grouped).
    sets: list of (qweight, scales, zeros, bias_or_None); returns the list of outputs [M, N_i]."""
    import ctypes
    tensors = [t for s in sets for t in s[:3]]
    _hip.need_gpu(x, *tensors)
    x = x.contiguous()
    M, K = x.shape
    n = len(sets)
    Ns = [int(s[0].shape[1]) for s in sets]
    ys = [torch.empty((M, Ni), dtype=x.dtype, device=x.device) for Ni in Ns]
    if M == 0:
        return ys
    L = _hip.lib()
    keep = [(s[1].contiguous(), s[2].contiguous()) for s in sets]
    arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
    Narr = (ctypes.c_int * n)(*Ns)
    need = L.bie_mpq_grouped_workspace_bytes(n, Narr, M, K, w_bit)
    ws = _hip.workspace(need, x.device)
    has_bias = any(s[3] is not None for s in sets)
    rc = L.bie_mpq_forward_grouped(_hip.ptr(x), n, arr([_hip.ptr(s[0]) for s in sets]), arr([_hip.ptr(k[0]) for k in keep]),
                                   arr([_hip.ptr(k[1]) for k in keep]), arr([_hip.ptr(s[3]) for s in sets]) if has_bias else None,
                                   arr([_hip.ptr(y) for y in ys]), Narr, _hip.ptr(ws), 0 if ws is None else ws.numel(),
                                   M, K, w_bit, group_size, int(bool(asym)), _hip.dt(x), _hip.stream())
    _hip.check(rc, "bie_mpq_forward_grouped")
    return ys


def mpq_forward(x, qweight, scales, qzeros, g_idx, a_bit, w_bit, asym):
    if a_bit != 16:
        raise RuntimeError(f"a_bit:{a_bit} has not been supported yet!")
    K = x.shape[1]
    return mpq_forward_impl(x, qweight.data, scales, qzeros, g_idx, w_bit, asym, _group_size(K, scales))


def mpq_dequant(qweight, scales, zeros, g_idx, w_bit, asym, group_size):
    _hip.need_gpu(qweight, scales, zeros)
    K = qweight.shape[0] * 32 // w_bit
    N = qweight.shape[1]
    out = torch.empty((K, N), dtype=scales.dtype, device=qweight.device)
    gptr = None if g_idx is None else g_idx.to(torch.int32).contiguous()
    rc = _hip.lib().bie_mpq_dequant(_hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                                    _hip.ptr(gptr), _hip.ptr(out), K, N, w_bit, group_size, int(bool(asym)),
                                    _hip.dt(scales), _hip.stream())
    _hip.check(rc, "bie_mpq_dequant")
    return out


def mpq_pack(weight, scales, zeros, g_idx, w_bit, asym, group_size):
    _hip.need_gpu(weight, scales, zeros)
    weight = weight.contiguous()
    K, N = weight.shape
    out = torch.empty((K * w_bit // 32, N), dtype=torch.int32, device=weight.device)
    gptr = None if g_idx is None else g_idx.to(torch.int32).contiguous()
    rc = _hip.lib().bie_mpq_pack(_hip.ptr(weight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                                 _hip.ptr(gptr), _hip.ptr(out), K, N, w_bit, group_size, int(bool(asym)),
                                 _hip.dt(weight), _hip.stream())
    _hip.check(rc, "bie_mpq_pack")
    return out


def mpq_grad_input(qweight, scales, qzeros, g_idx, grad_out, a_bit, w_bit, asym):
    _hip.need_gpu(qweight, scales, qzeros, grad_out)
    grad_out = grad_out.contiguous()
    M, N = grad_out.shape
    K = qweight.shape[0] * 32 // w_bit
    gs = _group_size(K, scales)
    gptr = None if gidx_is_trivial(g_idx, gs) else g_idx.to(torch.int32).contiguous()
    gx = torch.empty((M, K), dtype=grad_out.dtype, device=grad_out.device)
    rc = _hip.lib().bie_mpq_grad_input(_hip.ptr(grad_out), _hip.ptr(qweight), _hip.ptr(scales.contiguous()),
                                       _hip.ptr(qzeros.contiguous()), _hip.ptr(gptr), _hip.ptr(gx), M, K, N, w_bit, gs,
                                       int(bool(asym)), _hip.dt(grad_out), _hip.stream())
    _hip.check(rc, "bie_mpq_grad_input")
    return gx


# ---------------------------------------------------------------------------------------------- MBWQ
EXL2_ROWS_LEN = 20  # BIE_EXL2_ROWS_LEN (include/bie_hip.h)


def mbwq_trans_qweight(qweight, q_groups, use_mbw, height, groups, bits):
    """Returns (qweight, rows).  Mixed-bit layout: like the reference (mbwq_linear_cuda_kernel.cu:602-625) this is the
    load-time step that re-arranges the packed tensor IN PLACE for the kernels (its shuffle_kernel, :63-86 -- a no-op
    in the reference's build, exl2/config.h:16-21; here the half-pair layout of bie_mbwq_exl2_shuffle) and returns the
    band table: the reference's 7 ints followed by this library's group table (EXL2_ROWS_LEN ints in all; pass it on as
    it comes).  Like the reference's kernel launch it needs the tensor on the GPU, and like its blocking cudaMemcpy
    (:562) q_groups on the host.  Call it once per tensor."""
    import ctypes
    if use_mbw:
        t = qweight.data if hasattr(qweight, "data") else qweight
        _hip.need_gpu(t)
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError("mbwq_trans_qweight: qweight must be a contiguous int32 tensor")
        ver = qweight._version  # the caller's tensor / Parameter: `.data` carries a fresh counter that always reads 0
        if getattr(qweight, "_bie_exl2_shuffled", None) == (t.data_ptr(), ver):  # new contents (copy_, a new .data) may be prepared again
            raise RuntimeError("mbwq_trans_qweight: this qweight has already been re-arranged (a second pass would scramble it)")
        rows = (ctypes.c_int * EXL2_ROWS_LEN)()
        qg = q_groups.detach().to("cpu", torch.int16).contiguous()
        with torch.cuda.device(t.device):
            rc = _hip.lib().bie_mbwq_exl2_shuffle(_hip.ptr(t), qg.data_ptr(), groups, height, t.shape[1],
                                                  ctypes.cast(rows, ctypes.c_void_p), _hip.stream())
        _hip.check(rc, "bie_mbwq_exl2_shuffle")
        try:
            qweight._bie_exl2_shuffled = (t.data_ptr(), qweight._version)
        except AttributeError:
            pass
        return qweight, list(rows)
    if bits not in (2, 4):
        raise RuntimeError(f"Error: weight bit width:{bits} has not been supported yet!")
    return qweight, []


def mbwq_exl2_stream_copy(qweight, rows):
    """A COPY of a re-arranged exl2 tensor in the checkpoint's own form (the LSB-first chunk streams the reference stores and, its
    shuffle being a no-op, also saves after prepare_params): bie_mbwq_exl2_unshuffle on a clone.  What MBWQLinearCuda's state_dict
    hook writes, so that a saved checkpoint is the reference's format and `load -> prepare_params` re-arranges exactly once."""
    src = (qweight.data if hasattr(qweight, "data") else qweight).detach()
    home = src.device
    t = src.clone() if src.is_cuda else src.to("cuda")  # a prepared layer moved to the CPU for saving: the copy makes the trip (a kernel, as the shuffle was)
    _hip.need_gpu(t)
    keep, rp = _rows_arg(rows)
    rc = _hip.lib().bie_mbwq_exl2_unshuffle(_hip.ptr(t), rp, int(rows[5]), t.shape[1], _hip.stream())
    _hip.check(rc, "bie_mbwq_exl2_unshuffle")
    return t if home == t.device else t.to(home)


def mbwq_q42fp_weight(qweight, scales, zeros, group_size, bits, q_perm):
    _hip.need_gpu(qweight, scales, zeros)
    if scales.dtype != torch.float16:
        raise RuntimeError("mbwq_q42fp_weight: fp16 scales/zeros required")
    K = qweight.shape[0] * 32 // bits
    N = qweight.shape[1]
    out = torch.empty((K, N), dtype=torch.float16, device=qweight.device)
    perm = perm_or_none(q_perm)
    rc = _hip.lib().bie_mbwq_q4_dequant(_hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                                        _hip.ptr(perm), _hip.ptr(out), K, N, bits, group_size, _hip.stream())
    _hip.check(rc, "bie_mbwq_q4_dequant")
    return out


def _rows_arg(rows):
    import ctypes
    if rows is None or len(rows) != EXL2_ROWS_LEN:
        raise RuntimeError(f"exl2: the band table must be the {EXL2_ROWS_LEN}-int table mbwq_trans_qweight returned for THIS tensor "
                           f"(got {'none: the tensor was (re)loaded and not prepared' if rows is None else str(len(rows)) + ' ints'})")
    arr = (ctypes.c_int * EXL2_ROWS_LEN)(*[int(r) for r in rows])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


def mbwq_exl2fp_weight(qweight, scales, zeros, q_perm, q_group_map, rows):
    _hip.need_gpu(qweight, scales, zeros, q_perm, q_group_map)
    K = q_perm.shape[0]
    N = qweight.shape[1]
    out = torch.empty((K, N), dtype=torch.float16, device=qweight.device)
    keep, rp = _rows_arg(rows)
    rc = _hip.lib().bie_mbwq_exl2_dequant(_hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                                          _hip.ptr(q_perm), _hip.ptr(q_group_map), rp, _hip.ptr(out), K, N,
                                          scales.shape[0], _hip.stream())
    _hip.check(rc, "bie_mbwq_exl2_dequant")
    return out


def mbwq_q4_forward(x, qweight, scales, zeros, group_size, q_perm, bits):
    _hip.need_gpu(x, qweight, scales, zeros)
    if x.dtype != torch.float16:
        raise RuntimeError("mbwq_q4_forward: x must be torch.half")
    x = x.contiguous()
    M, K = x.shape
    if K != qweight.shape[0] * (32 // bits):
        raise RuntimeError("mbwq_q4_forward: x.size(1) != qweight.size(0) * (32 / bits)")
    N = qweight.shape[1]
    y = torch.empty((M, N), dtype=torch.float16, device=x.device)
    if M == 0:
        return y
    L = _hip.lib()
    ws = _hip.workspace(L.bie_mbwq_q4_workspace_bytes(M, K, N), x.device)  # (not the mixed-bit sizing: that one holds a dense weight image from M = 49)
    perm = perm_or_none(q_perm)
    rc = L.bie_mbwq_q4_forward(_hip.ptr(x), _hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                               _hip.ptr(perm), _hip.ptr(y), _hip.ptr(ws), 0 if ws is None else ws.numel(), M, K, N, bits,
                               group_size, _hip.stream())
    _hip.check(rc, "bie_mbwq_q4_forward")
    return y


# rows of x served by the streaming exl2 kernels (one pass over the packed weight): M <= 2 the decode kernel, 3 <= M <= 48 the same
# stream feeding v_mfma_f32_16x16x32_f16 (the reference keeps these in its fused kernel, exl2/q_gemm_kernel.cuh:90-549); beyond:
# the library's own prefill form (csrc/mbwq.hip: exl2_dequant_frag_kernel + mpq_dense_gemm_kernel; the switch is BIE_EXL2_DENSE_MIN_M there).  Measured at 4096x11008, 3/2-bit g32, random q_perm (profiles/r03_x_exl2_mfma.txt): M = 3...16
# 22.8-23.8 us, 32 39.3, 33 43.4, 64 59.2 against 48.1-51.4 us for reconstruct + library GEMM.  BIE_EXL2_MAX_M pins the switch for
# tools/ (8 = the round-2 split; the C-ABI itself takes M <= 64).
EXL2_GEMV_MAX_M = int(os.environ.get("BIE_EXL2_MAX_M", "48"))


def mbwq_exl2_forward(x, qweight, scales, zeros, q_perm, q_group_map, rows, use_cublas=False):
    _hip.need_gpu(x, qweight, scales, zeros, q_perm, q_group_map)
    if x.dtype != torch.float16:
        raise RuntimeError("mbwq_exl2_forward: x must be torch.half")
    x = x.contiguous()
    M = x.shape[0]
    K = q_perm.shape[0]
    N = qweight.shape[1]
    y = torch.empty((M, N), dtype=torch.float16, device=x.device)
    if M == 0:
        return y
    # One entry point for every M: M <= 2 the decode kernels, 3 <= M <= 48 the same stream on the matrix pipe, beyond that the prefill
    # form -- dequantise once into the MFMA fragment image, x[:, q_perm], dense MFMA GEMM (bie_mbwq_exl2_forward; the reference's
    # split is reconstruct + at::matmul, mbwq_linear_cuda_kernel.cu:968-1002, which `use_cublas` selects there: here the flag changes
    # nothing, there is no vendor GEMM on this path)
    L = _hip.lib()
    ws = _hip.workspace(L.bie_mbwq_workspace_bytes(M, K, N), x.device)
    keep, rp = _rows_arg(rows)
    rc = L.bie_mbwq_exl2_forward(_hip.ptr(x), _hip.ptr(qweight), _hip.ptr(scales.contiguous()), _hip.ptr(zeros.contiguous()),
                                 _hip.ptr(q_perm), _hip.ptr(q_group_map), rp, _hip.ptr(y), _hip.ptr(ws),
                                 0 if ws is None else ws.numel(), M, K, N, scales.shape[0], _hip.stream())
    _hip.check(rc, "bie_mbwq_exl2_forward")
    return y


EXL2_GROUP_MAX_ROWS = 48  # rows of x a grouped call takes: slabs of 16 rows (the kernel's range), one two-launch call per slab


def mbwq_exl2_forward_grouped(x, members):
    """Up to 8 exl2 layers on the SAME x [M, K] in two launches per 16 rows of x (bie_mbwq_exl2_forward_grouped; M <= 48: at 32 rows
    two grouped calls take 41 us for three 4096x4096 layers against 71 us for the members' own launches).  members: sequence of
    (qweight, scales, zeros, q_perm or None, q_group_map, rows) as `mbwq_exl2_forward` takes them.  Returns the outputs
    ([M, N_i] fp16, new tensors), or None when the set is outside the grouped range (irregular groups, K % 32, too many
    column blocks) -- the caller then runs the members one by one."""
    import ctypes
    _hip.need_gpu(x, *[t for m in members for t in m[:5] if t is not None])
    if x.dtype != torch.float16 or x.dim() != 2 or not 1 <= x.shape[0] <= EXL2_GROUP_MAX_ROWS or not 1 <= len(members) <= 8:
        return None
    x = x.contiguous()
    M, K = x.shape
    L = _hip.lib()
    arr = (_hip.Exl2ListEntry * len(members))()
    keep, outs = [], []
    for qweight, scales, zeros, q_perm, q_group_map, rows in members:
        tabl, rp = _rows_arg(rows)
        keep.append((tabl, rp, scales.contiguous(), zeros.contiguous()))
        outs.append(torch.empty((M, qweight.shape[1]), dtype=torch.float16, device=x.device))
    for r0 in range(0, M, 16):  # a slab of rows: x and every y are row-major, a slab of either is a contiguous block
        Ms = min(16, M - r0)
        for i, (qweight, scales, zeros, q_perm, q_group_map, rows) in enumerate(members):
            N = qweight.shape[1]
            _, rp, sc, ze = keep[i]
            arr[i] = _hip.Exl2ListEntry(None, qweight.data_ptr(), sc.data_ptr(), ze.data_ptr(), None if q_perm is None else q_perm.data_ptr(),
                                        q_group_map.data_ptr(), rp, outs[i].data_ptr() + r0 * N * 2, K, N, 0, 0)
        nbytes = L.bie_mbwq_exl2_grouped_workspace_bytes(len(members), arr, Ms)
        if nbytes == 0:
            return None
        ws = _hip.workspace(nbytes, x.device)
        rc = L.bie_mbwq_exl2_forward_grouped(x.data_ptr() + r0 * K * 2, Ms, len(members), arr, _hip.ptr(ws), ws.numel(), _hip.stream())
        _hip.check(rc, "bie_mbwq_exl2_forward_grouped")
    return outs
