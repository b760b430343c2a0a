"""Decode (M <= 2; W4: M <= 32 rows per launch, up to 64 rows as two row blocks) over a LIST of MPQ layers in ONE kernel launch per row
block (bie_mpq_list_*, include/bie_hip.h).

The reference launches one `quant_mm_kernel` per layer on the default stream
(layers/qlinear/nbit/cuda/mpq_layer.py:65 -> mpq_linear_cuda_kernel.cu:482-577).  A 4096x4096 W4 layer is 8.9 MB --
1.1 us of HBM time on MI355X, less than a kernel boundary -- so a per-layer launch can never be bandwidth-bound.
`MPQForwardList` hands the column tiles of many layers to one grid: independent entries (q/k/v, gate/up, the experts of an
MoE block, every layer of a speculative batch ...) and dependent chains (y of entry d is x of entry e: `depends_on`).

Results are those of `MPQLinearCuda.forward` entry by entry (same kernels' arithmetic, same roundings).
"""
import ctypes

import torch

from bitorch_engine import _hip


class MPQForwardList:
    """entries: sequence of dicts with keys x, qweight, scales, zeros, y and optionally bias, depends_on (index of an
    EARLIER entry whose `y` tensor IS this entry's `x`).  All tensors on one GPU; x [M, K], y [M, N] contiguous, dtype
    fp16 / bf16; the tensors' storage is frozen in the plan (update their CONTENTS, never rebind them).

    W4 with 32 < M <= 64 (a batch of decode streams): the rows are cut into two balanced blocks of <= 32, each a plan of its own over the
    row slices of every x / y (rows are independent; `depends_on` chains are an M <= 2 feature and are refused); forward() issues the
    blocks' launches back to back.  Every block streams the weights once; beyond 64 rows the per-layer GEMM is the better form and the
    constructor refuses."""

    def __new__(cls, entries, w_bit=4, group_size=128, asym=False):
        M = entries[0]["x"].reshape(-1, entries[0]["x"].shape[-1]).shape[0] if entries else 0
        if cls is MPQForwardList and w_bit == 4 and 32 < M <= 64:
            return _RowBlockedList(entries, w_bit, group_size, asym)
        return super().__new__(cls)

    def __init__(self, entries, w_bit=4, group_size=128, asym=False):
        if not entries:
            raise ValueError("MPQForwardList: empty list")
        L = _hip.lib()
        first = entries[0]
        dev = _hip.need_gpu(*[t for e in entries for t in (e["x"], e["qweight"], e["scales"], e["zeros"], e["y"], e.get("bias"))])
        self.M = first["x"].reshape(-1, first["x"].shape[-1]).shape[0]
        dtype = first["x"].dtype
        arr = (_hip.ListEntry * len(entries))()
        keep = []
        for i, e in enumerate(entries):
            x, y = e["x"], e["y"]
            K = x.shape[-1]
            N = e["qweight"].shape[1]
            if x.dtype != dtype or y.dtype != dtype or e["scales"].dtype != dtype:
                raise RuntimeError("MPQForwardList: one dtype per list")
            if not (x.is_contiguous() and y.is_contiguous() and e["qweight"].is_contiguous() and e["scales"].is_contiguous() and e["zeros"].is_contiguous()):
                raise RuntimeError("MPQForwardList: tensors must be contiguous")
            if x.numel() != self.M * K or y.numel() != self.M * N or e["qweight"].shape[0] != K * w_bit // 32:
                raise RuntimeError(f"MPQForwardList: entry {i} has inconsistent shapes")
            bias = e.get("bias")
            arr[i] = _hip.ListEntry(x.data_ptr(), e["qweight"].data_ptr(), e["scales"].data_ptr(), e["zeros"].data_ptr(),
                                    None if bias is None else bias.data_ptr(), y.data_ptr(), K, N, int(e.get("depends_on", -1)), 0)
            keep.append((x, e["qweight"], e["scales"], e["zeros"], bias, y))
        self._keep = keep  # the plan holds raw pointers
        self._entries = arr
        nbytes = L.bie_mpq_list_device_bytes(len(entries), arr, self.M, w_bit, group_size)
        if nbytes == 0:
            raise RuntimeError("MPQForwardList: this list is outside the one-launch decode range (w_bit 4: 1 <= M <= 32, groups of 32/64/128/256; "
                               "w_bit 2: M <= 2, groups of 64/128/256; K a multiple of one common group size)")
        self._mem = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        base = (self._mem.data_ptr() + 255) // 256 * 256
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _hip.check(L.bie_mpq_list_create(ctypes.byref(handle), len(entries), arr, self.M, w_bit, group_size, 1 if asym else 0,
                                             _hip.dt(first["x"]), base, nbytes), "bie_mpq_list_create")
        self._plan = handle
        self._L = L
        self.device = dev
        self.launches = L.bie_mpq_list_launches(handle)
        self.form = L.bie_mpq_list_form(handle)  # 0 lookup + FMA, 1 matrix pipe (K split inside a workgroup), 2 matrix pipe (x shared by a workgroup)

    def forward(self, stream=None):
        """Enqueue the launch on `stream` (a raw hipStream_t / None = the current stream of the plan's device)."""
        _hip.need_gpu(self._mem)
        st = _hip.stream() if stream is None else stream
        _hip.check(self._L.bie_mpq_list_forward(self._plan, st), "bie_mpq_list_forward")

    __call__ = forward

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            self._L.bie_mpq_list_destroy(plan)

    @classmethod
    def from_layers(cls, layers, xs, chain=False):
        """layers: prepared MPQLinearCuda modules (implicit g_idx); xs: one input tensor per layer, or ONE tensor when
        chain=True (layer i+1 reads layer i's output).  Returns (plan, ys)."""
        entries, ys = [], []
        l0 = layers[0]
        for i, layer in enumerate(layers):
            x = ys[-1] if (chain and i > 0) else (xs if (chain or isinstance(xs, torch.Tensor)) else xs[i])
            x2 = x.reshape(-1, x.shape[-1])
            y = torch.empty((x2.shape[0], layer.out_channels), dtype=x.dtype, device=x.device)
            zeros = layer.qzeros if layer.asym else layer.zeros
            entries.append({"x": x2, "qweight": layer.qweight.data, "scales": layer.scales, "zeros": zeros,
                            "bias": getattr(layer, "bias", None) if getattr(layer, "disable_bias", True) is False else None,
                            "y": y, "depends_on": i - 1 if (chain and i > 0) else -1})
            ys.append(y)
        return cls(entries, w_bit=l0.w_bit, group_size=l0.group_size, asym=l0.asym), ys


class _RowBlockedList:
    """MPQForwardList for 32 < M <= 64: one single-launch plan per block of <= 32 rows (see MPQForwardList)."""

    def __init__(self, entries, w_bit, group_size, asym):
        M = entries[0]["x"].reshape(-1, entries[0]["x"].shape[-1]).shape[0]
        nb = (M + 31) // 32
        cuts = [(M * b) // nb for b in range(nb + 1)]  # balanced blocks
        self.M = M
        self.blocks = []
        for r0, r1 in zip(cuts[:-1], cuts[1:]):
            sub = []
            for e in entries:
                d = dict(e)
                d["x"] = e["x"].reshape(M, -1)[r0:r1]
                d["y"] = e["y"].reshape(M, -1)[r0:r1]
                sub.append(d)
            plan = object.__new__(MPQForwardList)
            MPQForwardList.__init__(plan, sub, w_bit=w_bit, group_size=group_size, asym=asym)
            self.blocks.append(plan)
        self.device = self.blocks[0].device
        self.launches = sum(b.launches for b in self.blocks)

    def forward(self, stream=None):
        for b in self.blocks:
            b.forward(stream)

    __call__ = forward


class MBWQExl2ForwardList:
    """A list of exl2 (mixed 8/6/5/4/3/2-bit) decode layers in ONE launch (bie_mbwq_exl2_list_*): entry i is one
    `q_linear_cuda.mbwq_exl2_forward(x, qweight, scales, zeros, q_perm, q_group_map, rows)` call of the reference
    (mbwq_linear_cuda_kernel.cu:926-1007) with its own y.  entries: dicts with x, qweight, scales, zeros, q_perm (or None),
    q_group_map, rows (the band table from mbwq_trans_qweight, which also re-arranged qweight), y.  fp16, M <= 2 (M <= 16 with regular groups)."""

    def __init__(self, entries):
        L = _hip.lib()
        dev = _hip.need_gpu(*[t for e in entries for t in (e["x"], e["qweight"], e["scales"], e["zeros"], e["q_group_map"], e["y"], e.get("q_perm"))])
        self.M = entries[0]["x"].reshape(-1, entries[0]["x"].shape[-1]).shape[0]
        arr = (_hip.Exl2ListEntry * len(entries))()
        keep = []
        for i, e in enumerate(entries):
            x, y = e["x"], e["y"]
            if x.dtype != torch.float16 or y.dtype != torch.float16:
                raise RuntimeError("MBWQExl2ForwardList: fp16 only (as the reference kernels)")
            K, N = x.shape[-1], e["qweight"].shape[1]
            if len(e["rows"]) != 20:
                raise RuntimeError("MBWQExl2ForwardList: rows must be the 20-int table mbwq_trans_qweight returned")
            rows = (ctypes.c_int * 20)(*[int(v) for v in e["rows"]])
            perm = e.get("q_perm")
            for t in (x, y, e["qweight"], e["scales"], e["zeros"], e["q_group_map"]):
                if not t.is_contiguous():
                    raise RuntimeError("MBWQExl2ForwardList: tensors must be contiguous")
            arr[i] = _hip.Exl2ListEntry(x.data_ptr(), e["qweight"].data_ptr(), e["scales"].data_ptr(), e["zeros"].data_ptr(),
                                        None if perm is None else perm.data_ptr(), e["q_group_map"].data_ptr(),
                                        ctypes.cast(rows, ctypes.c_void_p), y.data_ptr(), K, N, 0, 0)
            keep.append((x, y, e["qweight"], e["scales"], e["zeros"], e["q_group_map"], perm, rows))
        self._keep = keep
        nbytes = L.bie_mbwq_exl2_list_device_bytes(len(entries), arr, self.M)
        if nbytes == 0:
            raise RuntimeError("MBWQExl2ForwardList: outside the one-launch decode range (1 <= M <= 2, or <= 16 with regular groups; K % 32 == 0)")
        self._mem = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        base = (self._mem.data_ptr() + 255) // 256 * 256
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _hip.check(L.bie_mbwq_exl2_list_create(ctypes.byref(handle), len(entries), arr, self.M, base, nbytes), "bie_mbwq_exl2_list_create")
        self._plan, self._L = handle, L

    def forward(self, stream=None):
        _hip.need_gpu(self._mem)
        _hip.check(self._L.bie_mbwq_exl2_list_forward(self._plan, _hip.stream() if stream is None else stream), "bie_mbwq_exl2_list_forward")

    __call__ = forward

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            self._L.bie_mbwq_exl2_list_destroy(plan)
