"""MPQLinearCuda: W{1,2,4,8}A16 linear on MI355X.  API mirror of reference
layers/qlinear/nbit/cuda/mpq_layer.py (Function :14-121, layer :124-224); the arithmetic is ONE fused
HIP kernel family for every M (decode GEMV, MFMA GEMM) instead of the reference's
"kernel for M<=32, else materialise the dense weight + cuBLAS" split (:59-65)."""
import math
import os
import typing

import torch
from torch.autograd import Function

from bitorch_engine.layers.qlinear.nbit import MPQLinearBase
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x

q_linear_cuda = import_extension("q_linear_cuda")

# counters of the automatic sibling grouping (tests and tools read them; reset with .clear())
GROUP_STATS = {"grouped_launches": 0, "served_from_group": 0, "single_launches": 0, "groups_confirmed": 0, "groups_dissolved": 0,
               "not_groupable": 0}
AUTO_GROUP = os.environ.get("BIE_AUTO_GROUP", "1") != "0"
GROUP_MAX_M = 16  # rows every grouped decode launch takes (bie_mpq_forward_grouped)
GROUP_MAX_M_WIDE = 32  # ... and the rows it takes for the sets bie_mpq_grouped_max_rows names (round 6: two row blocks per pass where that measured ahead)


def _set_rows_limit(members) -> int:
    """Rows of x up to which this confirmed set runs as one grouped launch: the library's answer for (K, total output columns, bit width, dtype)."""
    from bitorch_engine import _hip
    first = members[0]
    if first.scales.dtype not in (torch.float16, torch.bfloat16):
        return GROUP_MAX_M
    n_total = sum(int(m.out_channels) for m in members)
    return int(_hip.lib().bie_mpq_grouped_max_rows(int(first.in_channels), n_total, int(first.w_bit), _hip.F16 if first.scales.dtype == torch.float16 else _hip.BF16))


def _x_key(x):
    """Identity of an activation tensor for the sibling protocol: same storage, same version, same view -- and the same STREAM: a parked
    output is only handed to a sibling that is called on the stream the group launch was enqueued on (stream order is what makes it valid).
    An address names a tensor only while that tensor is ALIVE (the caching allocator hands a freed block, at version 0, to the next
    temporary of the same size): every place that stores a key stores the tensor beside it, so that no other tensor can come to carry
    the key while it is comparable (SiblingGroup.trace / .parked)."""
    stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
    return (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), x.dtype, x.device, stream)


class SiblingGroup:
    """MPQLinearCuda (or mixed-bit MBWQLinearCuda) children of ONE parent module that may consume the same activation (q/k/v, gate/up): what
    prepare_bie_layers() finds by structure and the first forward passes CONFIRM by observation, so that an unchanged caller

        q = self.q_proj(h); k = self.k_proj(h); v = self.v_proj(h)

    (reference call path: layers/qlinear/nbit/cuda/mpq_layer.py:206-224 -> one quant_mm_kernel launch per layer) runs ONE grouped
    decode launch (bie_mpq_forward_grouped).  A round = one forward of the parent, delimited by the member that is called first.  The
    first round is only observed: members that received the SAME tensor (storage pointer, version counter, shape, strides, stream)
    form a set, led by the one that was called first -- several sets per parent are fine (a flat block holds q/k/v AND gate/up), and
    o_proj, which sits beside q/k/v with the same shape but eats the attention output, ends up in none.  From then on a set's leader
    launches the whole set on its x (M <= 16 rows, eval mode) and parks the other members' outputs under the identity of x; a member
    called next with that tensor takes its parked output without launching anything; a member called with anything else, or before its
    leader, runs alone and leaves its set.  Nothing is assumed from names, and parked outputs nobody picks up for three rounds dissolve
    the group.

    The reference keeps no state between calls (mpq_layer.py:206-224), so this protocol must never be able to answer with another
    tensor's result: every trace / parked entry holds a STRONG REFERENCE to the x it was keyed on until the round ends, which pins the
    storage -- a caller doing `q_proj(h * a); k_proj(h * b)` gets two temporaries at two addresses, two keys, and no group.  (The cost:
    one round's inputs and parked outputs stay allocated until the parent's next forward.)"""

    def __init__(self, members):
        self.members = list(members)
        self.first = None          # the member called first in a forward of the parent: a call of it starts a new round
        self.sets = None           # confirmed: id(leader) -> [leader, member, ...]; None while observing
        self.leader_of = {}        # id(member) -> its set's leader
        self.trace = []            # (module, key, x) in call order, observation round only; x is held so its address cannot be recycled
        self.rounds_observed = 0
        self.parked = {}           # id(module) -> (key, x, output); output None = the leader could not group this call, run alone
        self.ungroupable = 0       # consecutive leader calls the grouped launch refused
        self.unclaimed = 0
        self.dead = False
        self._limits = {}          # id(leader) -> rows limit of its set

    def rows_limit(self, module) -> int:
        """How many rows of x a call of `module` may have and still go through the group: while observing, GROUP_MAX_M_WIDE (nothing is launched, and a caller that
        only ever decodes 17 .. 32 sequences must be observable too); confirmed: what the library takes for the module's set (cached per leader)."""
        if self.dead:
            return 0
        if self.sets is None:
            return GROUP_MAX_M_WIDE
        leader = self.leader_of.get(id(module))
        if leader is None:
            return GROUP_MAX_M  # in no set: the group still sees its calls (it may be the member that delimits the rounds)
        lim = self._limits.get(id(leader))
        if lim is None:
            try:
                lim = _set_rows_limit(self.sets[id(leader)]) if hasattr(leader, "w_bit") and hasattr(leader, "scales") and not getattr(leader, "use_mbw", False) else GROUP_MAX_M
            except Exception:
                lim = GROUP_MAX_M
            self._limits[id(leader)] = lim
        return lim

    def _dissolve(self):
        self.sets, self.leader_of, self.dead = None, {}, True
        self.parked.clear()
        GROUP_STATS["groups_dissolved"] += 1

    def _finish_round(self):
        if self.sets is None:
            if self.trace:
                self.rounds_observed += 1
                by_key = {}
                for m, k, _x in self.trace:
                    ms = by_key.setdefault(k, [])
                    if m not in ms:
                        ms.append(m)
                sets = {id(ms[0]): ms for ms in by_key.values() if len(ms) >= 2}
                if sets:
                    self.sets = sets
                    self.leader_of = {id(m): ms[0] for ms in sets.values() for m in ms}
                    GROUP_STATS["groups_confirmed"] += len(sets)
                elif self.rounds_observed >= 4:  # nobody shares an input here: stop looking
                    self.dead = True
            self.trace = []
        elif self.parked:
            self.unclaimed += 1
            if self.unclaimed >= 3:
                self._dissolve()
        self.parked.clear()

    def forward(self, module, x):
        """Returns the module's output, or None when the module has to run by itself."""
        if self.dead:
            return None
        key = _x_key(x)
        hit = self.parked.pop(id(module), None)
        if hit is not None and hit[0] == key:  # hit[1] is the leader's x, alive since the launch: the key cannot name another tensor
            self.unclaimed = 0
            if hit[2] is None:
                return None
            GROUP_STATS["served_from_group"] += 1
            return hit[2]
        if self.first is None:
            self.first = module
        if module is self.first:
            self._finish_round()
            if self.dead:
                return None
        if self.sets is None:
            self.trace.append((module, key, x))
            if len(self.trace) > 8 * len(self.members):  # the member that opened the round is not called once per forward: rounds cannot be told apart
                self.trace, self.dead = [], True
            return None
        leader = self.leader_of.get(id(module))
        if leader is module:
            members = self.sets[id(module)]
            outs = type(module)._grouped_or_none(members, x)
            if outs is None:  # not a call the grouped launch takes (dtype, rows, explicit g_idx ...): everybody runs alone, nobody is evicted
                GROUP_STATS["not_groupable"] += 1
                self.ungroupable += 1
                if self.ungroupable >= 3:
                    self._dissolve()
                    return None
                for m in members[1:]:
                    self.parked[id(m)] = (key, x, None)
                return None
            self.ungroupable = 0
            GROUP_STATS["grouped_launches"] += 1
            for m, o in zip(members[1:], outs[1:]):
                self.parked[id(m)] = (key, x, o)
            return outs[0]
        if leader is not None:  # launched with its leader's x but asked for another tensor (or called before its leader): not a sibling after all
            members = self.sets[id(leader)]
            members.remove(module)
            self._limits.pop(id(leader), None)
            del self.leader_of[id(module)]
            if len(members) < 2:
                del self.sets[id(leader)]
                self.leader_of.pop(id(leader), None)
                if not self.sets:
                    self._dissolve()
        return None


def find_sibling_groups(model: torch.nn.Module) -> int:
    """Attach a SiblingGroup to every set of >= 2 children of one parent that could run as one grouped decode launch: MPQLinearCuda
    layers sharing in_channels / w_bit / group_size / asym / dtype, and mixed-bit (exl2) MBWQLinearCuda layers sharing in_channels
    (called by utils.model_helper.prepare_bie_layers).  Returns the number of candidate groups."""
    from .mbwq_layer import MBWQLinearCuda
    n = 0
    for parent in model.modules():
        buckets = {}
        for c in parent.children():
            if isinstance(c, MPQLinearCuda):
                if c.w_bit in (4, 2):
                    buckets.setdefault(("mpq", c.in_channels, c.w_bit, c.group_size, bool(c.asym), c.dtype), []).append(c)
            elif isinstance(c, MBWQLinearCuda) and c.use_mbw:
                buckets.setdefault(("exl2", c.in_channels), []).append(c)
        for members in buckets.values():
            if len(members) >= 2:
                g = SiblingGroup(members)
                for m in members:
                    m._bie_group = g
                n += 1
    return n


def clear_sibling_groups(model: torch.nn.Module) -> int:
    """Detach every SiblingGroup below `model` (prepare_bie_layers(model, group_siblings=False)): each layer launches for itself again,
    exactly the reference's call pattern (mpq_layer.py:206-224).  Returns the number of layers that were in a group."""
    n = 0
    for m in model.modules():
        if getattr(m, "_bie_group", None) is not None:
            m._bie_group = None
            n += 1
    return n


class MPQLinearCudaFunction(Function):
    @staticmethod
    def forward(ctx, x, qweight, a_bit, w_bit, scales, zeros, g_idx, asym, is_training, privileged_grad=None):
        x2, lead = flatten_x(x)
        out = q_linear_cuda.mpq_forward(x2, qweight, scales, zeros, g_idx, a_bit, w_bit, asym)
        if is_training:
            qweight.privileged_grad = privileged_grad
            qweight.scales, qweight.zeros, qweight.g_idx = scales, zeros, g_idx
            qweight.w_bit, qweight.asym, qweight.layer_type = w_bit, asym, 1
            ctx.a_bit = a_bit
            ctx.save_for_backward(x2, qweight)
        return unflatten_x(out, lead)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy, lead = flatten_x(output_gradient)
        x2, qweight = ctx.saved_tensors
        gy = gy.to(x2.dtype)
        gx = q_linear_cuda.mpq_grad_input(qweight.data, qweight.scales, qweight.zeros, qweight.g_idx, gy, ctx.a_bit,
                                          qweight.w_bit, qweight.asym)
        if qweight.requires_grad:
            qweight.privileged_grad = x2.t().mm(gy)
        return (unflatten_x(gx, lead),) + (None,) * 9


class MPQLinearCuda(MPQLinearBase):
    """Mixed-precision quantised linear layer (weights 1/2/4/8 bit, activations fp16/bf16)."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.qweight.layer_type = 1
        self._gidx_trivial = None
        self._bie_group = None  # SiblingGroup, set by prepare_bie_layers (find_sibling_groups)
        self.check_parameters()

    def check_parameters(self) -> None:
        assert self.w_bit in [1, 2, 4, 8], f"The value of w_bit ({self.w_bit}) must be 1, 2, 4 or 8."
        assert self.a_bit == 16, f"The value of a_bit ({self.a_bit}) must be 16."

    def prepare_params(self) -> None:
        """Decode the double-quantised statistics into per-group `scales` / `zeros` ([G, N], layer dtype) and
        drop the load-only buffers -- reference mpq_layer.py:163-204.  Elementwise torch ops in the layer
        dtype (each op rounds once, like the reference); runs once per layer, on whatever device the
        buffers live on."""
        try:
            if self.use_gba_quant:
                if self.group_size < 256:  # larger groups are stored without double quantisation
                    shape = (math.ceil(self.in_channels / self.group_size), self.out_channels)
                    if self.asym:
                        codes = self.qscales.unsqueeze(-1) if self.w_bit == 2 else self.qscales
                        self.zeros = self.qzeros
                    else:
                        stat = self.qstatistic.to(torch.uint8)
                        codes = stat >> 4
                        zcodes = stat & 0x0F
                        self.zeros = ((zcodes.to(self.dtype) - self.qzeros_zeros) * self.qzeros_scales).view(shape)
                    self.scales = ((codes.to(self.dtype) - self.qscales_zeros) * self.qscales_scales).view(shape)
                for name in ("qscales_zeros", "qscales_scales") + (("qscales",) if self.asym else
                                                                     ("qstatistic", "qzeros_zeros", "qzeros_scales")):
                    delattr(self, name)
            else:
                self.zeros = self.qzeros
            if self.disable_bias:
                del self.bias
            del self.wf
            self._gidx_trivial = None
        except Exception as e:
            raise RuntimeError(f"Error occurred during parameter preparation in MPQLinearCuda layer: {e}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dev = x.device
        if not all(t.device == dev for t in (self.qweight, self.scales, self.zeros, self.g_idx)):
            raise RuntimeError("Some tensors are not on the correct device, please make sure to move the layer to "
                               "the correct device and call 'finalize_quantized_layers'.")
        if self.training or torch.is_grad_enabled() and x.requires_grad:
            out = MPQLinearCudaFunction.apply(x, self.qweight, self.a_bit, self.w_bit, self.scales, self.zeros,
                                              self.g_idx, self.asym, self.training, self.privileged_grad)
            return out if self.disable_bias else out + self.bias
        # inference fast path: bias fused into the kernel epilogue; whether g_idx is the trivial k // group_size is
        # remembered on the g_idx tensor itself (keyed by its version), so load_state_dict / in-place edits invalidate it
        x2, lead = flatten_x(x)
        if AUTO_GROUP and self._bie_group is not None and 0 < x2.shape[0] <= GROUP_MAX_M_WIDE and x2.shape[0] <= self._bie_group.rows_limit(self):
            out = self._bie_group.forward(self, x)  # decode: siblings that share x run as ONE grouped launch (see SiblingGroup)
            if out is not None:
                return out
        GROUP_STATS["single_launches"] += 1
        out = q_linear_cuda.mpq_forward_impl(x2, self.qweight.data, self.scales, self.zeros, self.g_idx, self.w_bit,
                                             self.asym, self.group_size, None if self.disable_bias else self.bias)
        return unflatten_x(out, lead)

    @staticmethod
    def forward_grouped(layers: typing.Sequence["MPQLinearCuda"], x: torch.Tensor) -> typing.List[torch.Tensor]:
        """Several layers that consume the SAME activation (q/k/v, gate/up of a transformer block) in ONE decode launch
        (bie_mpq_forward_grouped): the reference launches `quant_mm_kernel` once per layer (mpq_layer.py:65); at M <= 16 a
        launch of this size is mostly fixed cost, and three 4096x4096 projections in one grid take 10.1 us instead of 3 x 6.0.
        Falls back to the layers' own forward when the set is not groupable (training, different bit widths / group sizes /
        dtypes, explicit g_idx, more than 16 rows or 8 layers)."""
        outs = MPQLinearCuda._grouped_or_none(layers, x)
        return outs if outs is not None else [l(x) for l in layers]

    @staticmethod
    def _grouped_or_none(layers, x):
        """The grouped launch, or None when this call is not one it takes."""
        first = layers[0]
        x2, lead = flatten_x(x)
        same = all(l.w_bit == first.w_bit and l.group_size == first.group_size and l.asym == first.asym and l.in_channels == first.in_channels
                   and l.scales.dtype == first.scales.dtype and not l.training for l in layers)
        max_rows = (_set_rows_limit(layers) if same and first.w_bit == 4 else 16) if first.w_bit == 4 else 2  # W4: 16 rows, 32 for the sets the library names (two row blocks); W2: 2 (pair lookup)
        ok = (same and first.w_bit in (4, 2) and 1 <= x2.shape[0] <= max_rows and 2 <= len(layers) <= 8 and x2.dtype == first.scales.dtype
              and not (torch.is_grad_enabled() and x.requires_grad)
              and all(q_linear_cuda.gidx_is_trivial(l.g_idx, l.group_size) for l in layers))
        if not ok:
            return None
        sets = [(l.qweight.data, l.scales, l.zeros, None if l.disable_bias else l.bias) for l in layers]
        outs = q_linear_cuda.mpq_forward_grouped_impl(x2, sets, first.w_bit, first.asym, first.group_size)
        return [unflatten_x(o, lead) for o in outs]

