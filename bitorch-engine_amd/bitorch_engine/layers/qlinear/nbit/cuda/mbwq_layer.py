"""MBWQLinearCuda: uniform 4/2-bit (GPTQ-like) and mixed 8/6/5/4/3/2-bit (exl2 layout) fp16 linear.
API mirror of reference layers/qlinear/nbit/cuda/mbwq_layer.py (Function :14-122, layer :125-372)."""
import math
import typing

import torch
from torch.autograd import Function

from bitorch_engine.layers.qlinear.nbit import MPQLinearBase, MPQWeightParameter
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x
from .utils import unpack_qweight, make_group_map

q_linear_cuda = import_extension("q_linear_cuda")


class MBWQLinearCudaFunction(Function):
    @staticmethod
    def forward(ctx, x, qweight, use_mbw, is_train, scales, zeros, group_size, q_perm=None, bits=4,
                privileged_grad=None, q_group_map=None, rows=None):
        x2, lead = flatten_x(x)
        if use_mbw:
            out = q_linear_cuda.mbwq_exl2_forward(x2, qweight, scales, zeros, q_perm, q_group_map, rows, False)
        else:
            out = q_linear_cuda.mbwq_q4_forward(x2, qweight, scales, zeros, group_size, q_perm, bits)
        if is_train:
            qweight.scales, qweight.zeros, qweight.q_perm = scales, zeros, q_perm
            qweight.privileged_grad, qweight.group_size = privileged_grad, group_size
            qweight.q_group_map, qweight.rows = q_group_map, rows
            qweight.layer_type, qweight.w_bit, qweight.asym, qweight.g_idx = 2, bits, False, None
            ctx.save_for_backward(x2, qweight)
        return unflatten_x(out, lead)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy, lead = flatten_x(output_gradient)
        x2, qweight = ctx.saved_tensors
        gy = gy.to(x2.dtype)
        gx = gy.mm(unpack_qweight(qweight).to(x2.dtype).t())
        if qweight.requires_grad:
            qweight.privileged_grad = x2.t().mm(gy)
        return (unflatten_x(gx, lead),) + (None,) * 11


class MBWQLinearCuda(MPQLinearBase):
    def __init__(self, *args, use_mbw: bool = True, groups=64, rows_packed=64, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.qweight.layer_type = 2
        self.use_mbw, self.groups, self.rows_packed = use_mbw, groups, rows_packed
        self.rows = [0] * 7  # rows_8, rows_6, rows_5, rows_4, rows_3, rows_2 (cumulative k), kernel_p bit mask (prepare_params: + the group table)
        self._bie_group = None  # SiblingGroup, set by prepare_bie_layers (mpq_layer.find_sibling_groups)
        self._exl2_mark = None  # (data_ptr, _version) of qweight as prepare_params left it (the private half-pair layout); see _exl2_current
        self.check_parameters()
        self._register_state_dict_hook(MBWQLinearCuda._checkpoint_format_hook)

    # ---- the private layout never leaves the process ---------------------------------------------------------------------------------
    # prepare_params() re-arranges the mixed-bit qweight in place for the kernels (q_linear_cuda.mbwq_trans_qweight; the reference's own
    # shuffle hook is a no-op, exl2/config.h:16-21, so ITS tensor is the checkpoint's stream before and after).  Three rules keep the
    # on-disk contract and make a stale tensor fail loudly instead of multiplying garbage:
    #   * state_dict() writes qweight BACK in the stream form (a copy; bie_mbwq_exl2_unshuffle): a saved checkpoint is the reference's;
    #   * tensors that arrive through load_state_dict / a new .data are streams: the mark no longer matches, forward() refuses until
    #     prepare_params() has run again (which re-arranges exactly once);
    #   * .to() / .cuda() move the bytes as they are: _apply carries the mark over.
    def _exl2_current(self) -> bool:
        # the PARAMETER's version counter (`.data` hands out a fresh counter that always reads 0); writes through `.data` / raw pointers
        # are invisible to it, as they are to autograd
        return self._exl2_mark == (self.qweight.data_ptr(), self.qweight._version)

    def _require_prepared(self):
        if self.use_mbw and not self._exl2_current():
            raise RuntimeError("MBWQLinearCuda: qweight changed since prepare_params() (load_state_dict, a new .data, or never prepared): its "
                               "contents are the checkpoint's chunk streams, not the layout the kernels read; call prepare_params() again")

    def _apply(self, fn, *args, **kwargs):
        was = self.use_mbw and self._exl2_mark is not None and self._exl2_current()
        out = super()._apply(fn, *args, **kwargs)
        if was:
            self._exl2_mark = (self.qweight.data_ptr(), self.qweight._version)
            try:
                self.qweight._bie_exl2_shuffled = self._exl2_mark
            except AttributeError:
                pass
        return out

    def __getstate__(self):
        # copy.deepcopy / pickle: the copy's qweight lives at a new address, so the (address, version) mark cannot travel as it is; what travels is
        # WHETHER the layer was in its prepared state, and __setstate__ re-keys the mark on the copy's own tensor
        state = self.__dict__.copy()
        state["_exl2_prepared_at_copy"] = bool(self.use_mbw and self._exl2_mark is not None and self._exl2_current())
        return state

    def __setstate__(self, state):
        was = state.pop("_exl2_prepared_at_copy", False)
        super().__setstate__(state)
        self._exl2_mark = (self.qweight.data_ptr(), self.qweight._version) if was else None
        if was:
            try:
                self.qweight._bie_exl2_shuffled = self._exl2_mark
            except AttributeError:
                pass

    @staticmethod
    def _checkpoint_format_hook(module, state_dict, prefix, local_metadata):
        key = prefix + "qweight"
        if module.use_mbw and key in state_dict and module._exl2_mark is not None and module._exl2_current():
            state_dict[key] = q_linear_cuda.mbwq_exl2_stream_copy(module.qweight, module.rows)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        before = (self.qweight.data_ptr(), self.qweight._version) if self.use_mbw else None
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        # forget only when the base class really wrote the tensor: a rejected load (shape mismatch -> error_msgs) leaves a prepared
        # layer exactly as usable as it was
        if self.use_mbw and prefix + "qweight" in state_dict and (self.qweight.data_ptr(), self.qweight._version) != before:
            self._forget_layout(keep_group_map=prefix + "q_group_map" in state_dict)

    def _forget_layout(self, keep_group_map: bool = False):
        """qweight holds checkpoint streams again: no table of the private layout may stay attached to it (unpack_qweight and the static
        *fp_weight helpers read qweight.rows): until prepare_params() runs, every exl2 call on this tensor fails on the table.
        The group map goes too: it is a function of q_groups, and a second checkpoint with the same tensor shapes but another
        bit allocation would otherwise get new rows and a new shuffle with the OLD map -- scales and zeros of the wrong groups, silently.
        It stays only when the checkpoint itself supplied one."""
        self._exl2_mark = None
        self.rows = [0] * 7
        self.qweight.rows = None
        if not keep_group_map:
            self.q_group_map = None
            self.qweight.q_group_map = None

    def check_parameters(self) -> None:
        assert self.dtype == torch.half, f"The value of dtype ({self.dtype}) must be torch.half."
        self.register_buffer("q_perm", torch.zeros(self.in_channels, dtype=torch.short))
        self.register_buffer("channel_scale", torch.ones((1, 1, self.in_channels), dtype=self.dtype))
        if not self.use_mbw:
            assert self.w_bit in [2, 4], f"The value of w_bit ({self.w_bit}) must be 4 or 2."
            assert self.group_size >= 32, f"The value of group_size ({self.group_size}) must >= 32."
            return
        shape = (math.ceil(self.groups), math.ceil(self.out_channels))
        self.qweight = MPQWeightParameter(torch.empty((self.rows_packed, self.out_channels), dtype=torch.int32),
                                          requires_grad=False, layer_type=2)
        self.register_buffer("q_groups", torch.empty(self.groups * 2, dtype=torch.short))
        self.register_buffer("zeros", torch.empty(shape, dtype=self.dtype))
        self.register_buffer("scales", torch.empty(shape, dtype=self.dtype))
        self.q_group_map = None

    def load_state_dict(self, state_dict, strict=True) -> None:
        """Like the reference (:205-237): exl2 tensors whose shapes differ from the constructor's guess are
        adopted as they come."""
        # the layer's own tensors (detach() shares the version counter), collected WITHOUT state_dict(): its checkpoint hook would hand out (and
        # compute, on the GPU) a stream copy of a prepared qweight
        own = {n: p.detach() for n, p in self._parameters.items() if p is not None}
        own.update({n: b for n, b in self._buffers.items() if b is not None and n not in self._non_persistent_buffers_set})
        for name, value in state_dict.items():
            if name not in own:
                if strict:
                    raise KeyError(f"Missing key {name} in own state")
                continue
            if own[name].shape == value.shape:
                own[name].copy_(value)
            elif name in ("scales", "zeros", "q_perm", "q_groups", "q_group_map", "qweight"):
                print(f"Warning: Shape mismatch for: {name}, expected: {own[name].shape}, got: {value.shape}. "
                      f"Use the value in state_dict directly.")
                target = self._parameters.get(name)
                if target is None:
                    target = self._buffers.get(name)
                # state_dict() hands out detached aliases: re-pointing one of THOSE (what the reference's loop does, :228-231) leaves
                # the module's own tensor untouched; adopt the value on the registered parameter / buffer itself
                (own[name] if target is None else target).data = value.data
            if name == "qweight":
                self._forget_layout(keep_group_map="q_group_map" in state_dict)  # the checkpoint's streams: prepare_params() has to run (again)
        if not strict:
            missing = set(own.keys()) - set(state_dict.keys())
            if missing:
                print(f"Warning: Missing keys in state_dict: {missing}")

    def set_scales(self, scales: torch.Tensor = None) -> None:
        self.scales = scales
        self.qweight.scales = scales

    def set_zeros(self, zeros: torch.Tensor = None) -> None:
        self.zeros = zeros
        self.qweight.zeros = zeros

    def prepare_params(self) -> None:
        try:
            self.qweight.scales, self.qweight.zeros, self.qweight.q_perm = self.scales, self.zeros, self.q_perm
            height, groups = self.q_perm.size(0), self.scales.size(0)
            if self.use_mbw:
                if self._exl2_mark is not None and self._exl2_current():
                    raise RuntimeError("this layer's qweight has already been re-arranged (a second pass would scramble it)")
                self.qweight.data, self.rows = q_linear_cuda.mbwq_trans_qweight(self.qweight, self.q_groups, True,
                                                                                height, groups, self.w_bit)
                self._exl2_mark = (self.qweight.data_ptr(), self.qweight._version)
                if self.q_group_map is None:
                    self.q_group_map = make_group_map(self.q_groups, self.qweight.shape[0])
                self.qweight.q_group_map, self.qweight.rows = self.q_group_map, self.rows
            else:
                q_linear_cuda.mbwq_trans_qweight(self.qweight, None, False, height, groups, self.w_bit)
            for name in ("qzeros_zeros", "qzeros_scales", "qscales_zeros", "qscales_scales", "qstatistic"):
                if hasattr(self, name):
                    delattr(self, name)
            for name in (("bias",) if self.disable_bias else ()) + ("wf", "g_idx"):
                if hasattr(self, name):  # a second prepare_params() (after a checkpoint was loaded into a prepared layer) finds them gone
                    delattr(self, name)
        except Exception as e:
            raise RuntimeError(f"Error occurred during parameter preparation in MBWQLinearCuda layer: {e}")

    @staticmethod
    def q42fp_weight(qweight, scales, zeros, group_size, bits, q_perm) -> torch.Tensor:
        return q_linear_cuda.mbwq_q42fp_weight(qweight, scales, zeros, group_size, bits, q_perm)

    @staticmethod
    def exl2fp_weight(qweight, scales, zeros, q_perm, q_group_map, rows) -> torch.Tensor:
        return q_linear_cuda.mbwq_exl2fp_weight(qweight, scales, zeros, q_perm, q_group_map, rows)

    def _channel_scale_is_one(self) -> bool:
        """Memoised on the tensor's identity and version counter (a blocking read, once)."""
        cs = self.channel_scale
        key = (cs.data_ptr(), cs._version, tuple(cs.shape))
        if getattr(self, "_cs_one_key", None) != key:
            if cs.is_cuda and torch.cuda.is_current_stream_capturing():
                return False  # no host read under capture: this call multiplies (the reference's path); the next uncaptured call settles it
            self._cs_one_key, self._cs_one = key, bool((cs == 1).all().item())
        return self._cs_one

    @staticmethod
    def forward_grouped(layers: typing.Sequence["MBWQLinearCuda"], x: torch.Tensor) -> typing.List[torch.Tensor]:
        """Mixed-bit layers that consume the SAME one-row activation (q/k/v, gate/up) in two launches instead of one
        gemm_half_q_half_kernel launch per layer (mbwq_linear_cuda_kernel.cu:926-1007): bie_mbwq_exl2_forward_grouped.  Groupable:
        eval mode, up to 48 rows of fp16 x in slabs of sixteen (measured: 3 x 4096x4096 at 8 / 16 / 32 rows 14.0 / 20.7 / 41 us grouped against 35 / 38 / 71 us alone), every channel_scale all ones (x * 1 is x: the members do share their input), regular groups.
        Anything else runs the members one by one."""
        outs = MBWQLinearCuda._grouped_or_none(layers, x)
        return outs if outs is not None else [l(x) for l in layers]

    @staticmethod
    def _grouped_or_none(layers, x):
        """The grouped call, or None when this call is not one it takes (mpq_layer.SiblingGroup then lets every member run alone)."""
        x2, lead = flatten_x(x)
        for l in layers:
            l._require_prepared()
        ok = (2 <= len(layers) <= 8 and 1 <= x2.shape[0] <= q_linear_cuda.EXL2_GROUP_MAX_ROWS and x2.dtype == torch.half and not (torch.is_grad_enabled() and x.requires_grad)
              and all(l.use_mbw and not l.training and l.q_group_map is not None and l.in_channels == layers[0].in_channels
                      and l._channel_scale_is_one() for l in layers))
        outs = None
        if ok:
            outs = q_linear_cuda.mbwq_exl2_forward_grouped(x2, [(l.qweight.data, l.scales, l.zeros, l.q_perm, l.q_group_map, l.rows) for l in layers])
        if outs is None:
            return None
        return [unflatten_x(o if l.disable_bias else o + l.bias, lead) for l, o in zip(layers, outs)]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .mpq_layer import AUTO_GROUP
        if AUTO_GROUP and getattr(self, "_bie_group", None) is not None and self.use_mbw and not self.training and 0 < x.numel() <= q_linear_cuda.EXL2_GROUP_MAX_ROWS * x.shape[-1]:
            out = self._bie_group.forward(self, x)  # decode: siblings that share x run as one grouped call (mpq_layer.SiblingGroup)
            if out is not None:
                return out
        return self._forward_alone(x)

    def _forward_alone(self, x: torch.Tensor) -> torch.Tensor:
        self._require_prepared()
        if not (self.use_mbw and not self.training and self._channel_scale_is_one()):  # x * 1: one elementwise launch per call for nothing
            x = x.mul(self.channel_scale)
        extra = (self.q_group_map, self.rows) if self.use_mbw else ()
        out = MBWQLinearCudaFunction.apply(x, self.qweight, self.use_mbw, self.training, self.scales, self.zeros,
                                           self.group_size, self.q_perm, self.w_bit, self.privileged_grad, *extra)
        return out if self.disable_bias else out + self.bias
