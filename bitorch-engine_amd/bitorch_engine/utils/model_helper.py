"""Helpers either side of the hot path, mirroring reference utils/model_helper.py: flatten_x / unflatten_x (:10-51),
pad_embedding_dim (:54-82), pad_last_2_dims_to_multiple_of_128 (:85-117), binary_matmul_forward_post_processing (:120-155),
prepare_bie_layers (:158-196), pack_bie_layers / save_checkpoint / load_checkpoint (:199-283), init_weight (:286-327)."""
import math
from typing import Optional, Tuple, Type

import torch
import torch.nn.functional as F


def flatten_x(x: torch.Tensor):
    """[..., K] -> ([M, K], leading shape)."""
    lead = list(x.shape[:-1])
    return x.reshape(-1, x.shape[-1]), lead


def unflatten_x(x: torch.Tensor, shape: list):
    return x.view(shape + [x.shape[-1]])


def pad_embedding_dim(weight: torch.Tensor) -> torch.Tensor:
    """Pad the embedding dimension (dim 1) up to a multiple of 8 with -1 (a cleared bit once sign-packed) --
    reference utils/model_helper.py:54-82."""
    extra = (-weight.shape[1]) % 8
    if extra == 0:
        return weight
    fill = torch.full((weight.shape[0], extra), -1.0, dtype=torch.float, device=weight.device)
    return torch.cat([weight, fill], dim=1)


def pad_last_2_dims_to_multiple_of_128(tensor: torch.Tensor):
    """(padded tensor, rows added to dim -2): the last two dims rounded up to multiples of 128, padded with ZEROS on the
    right / bottom -- reference utils/model_helper.py:85-117 (it returns the pair although it is annotated -> Tensor)."""
    rows, cols = tensor.shape[-2], tensor.shape[-1]
    add_rows, add_cols = (-rows) % 128, (-cols) % 128
    if add_rows or add_cols:
        tensor = F.pad(tensor, (0, add_cols, 0, add_rows), mode="constant", value=0)
    return tensor, add_rows


def binary_matmul_forward_post_processing(tensor: torch.Tensor, shape_pre: list, x_pad_sec_last: int, y_pad_sec_last: int,
                                          k: int) -> torch.Tensor:
    """Undo the padding of a batched XOR-popcount product [B, m_pad, n_pad]: drop the padded rows / columns, restore the
    leading dims `shape_pre`, and map the popcount to the +-1 dot product k - 2*popc -- reference :120-155."""
    m_keep = tensor.shape[-2] - x_pad_sec_last
    n_keep = tensor.shape[-1] - y_pad_sec_last
    tensor = tensor[:, :m_keep, :n_keep]
    return k - 2 * tensor.reshape(list(shape_pre) + [m_keep, n_keep])


def _bie_layer_kinds(with_embedding: bool):
    from bitorch_engine.layers.qlinear.nbit import MPQLinearBase, nBitLinearBase
    from bitorch_engine.layers.qconv.nbit import nBitConv2dBase
    from bitorch_engine.layers.qlinear.binary import BinaryLinearBase
    from bitorch_engine.layers.qconv.binary import BinaryConv2dBase
    kinds = [BinaryConv2dBase, nBitConv2dBase, BinaryLinearBase, nBitLinearBase, MPQLinearBase]
    if with_embedding:
        from bitorch_engine.layers.qembedding.binary import BinaryEmbeddingCuda
        kinds.append(BinaryEmbeddingCuda)
    return tuple(kinds)


def prepare_bie_layers(model: torch.nn.Module, layers=None, group_siblings: Optional[bool] = None) -> None:
    """Call prepare_params() on every BIE layer below `model` (decode double-quantised statistics, build band tables, ...);
    `layers` optionally restricts the layer classes -- reference :158-196.

    group_siblings (this library only; the reference keeps no state between calls, layers/qlinear/nbit/cuda/mpq_layer.py:206-224):
    True / None-with-BIE_AUTO_GROUP!=0 registers sibling layers that may share an input (q/k/v, gate/up) as candidate groups, so that an
    unchanged module tree runs ONE grouped decode launch per set; False leaves every layer a launch of its own AND removes groups an
    earlier call attached.  Grouping relies on one contract the reference does not need -- between the first and the last sibling call of
    one parent forward, x must not be overwritten through `.data`, a raw pointer or any writer autograd's version counter does not see
    ("Contract differences") -- pass False where that cannot be promised."""
    kinds = tuple(layers) if layers else _bie_layer_kinds(True)
    for i, module in enumerate(model.modules()):
        if i > 0 and isinstance(module, kinds):
            module.prepare_params()
    # MI355X side of the same hook: MPQLinearCuda siblings that may share an input (q/k/v, gate/up) are registered as candidate groups;
    # their first forward passes confirm which of them really receive the same tensor, and from then on ONE grouped decode launch
    # serves them (layers/qlinear/nbit/cuda/mpq_layer.py::SiblingGroup).  The caller's code does not change.
    try:
        from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    except ImportError:
        return
    if group_siblings is None:
        group_siblings = mpq_layer.AUTO_GROUP
    if group_siblings:
        mpq_layer.find_sibling_groups(model)
    else:
        mpq_layer.clear_sibling_groups(model)


def pack_bie_layers(model: torch.nn.Module, qweight_only: bool = True, layers=None) -> None:
    """generate_quantized_weight(qweight_only) on every BIE layer below `model`: bit-pack the weights before torch.save()
    (qweight_only drops the unpacked training weights) -- reference :199-233."""
    kinds = tuple(layers) if layers else _bie_layer_kinds(False)
    for i, module in enumerate(model.modules()):
        if i > 0 and isinstance(module, kinds):
            module.generate_quantized_weight(qweight_only=qweight_only)


def save_checkpoint(model: torch.nn.Module, name: str, qweight_only: bool = True) -> None:
    """Pack the quantised layers, then torch.save({'state_dict': ...}) -- same file layout as reference :236-262."""
    pack_bie_layers(model, qweight_only)
    torch.save({"state_dict": model.state_dict()}, name)


def load_checkpoint(model: torch.nn.Module, checkpoint_path: str, qweight_only: bool = True) -> None:
    """Pack the model's layers (so the packed buffers exist with the right shapes), then load a save_checkpoint() file
    non-strictly -- reference :265-283."""
    pack_bie_layers(model, qweight_only)
    state = torch.load(checkpoint_path, map_location="cpu")
    model.load_state_dict(state["state_dict"], strict=False)


def init_weight(weight: torch.Tensor, cls: Type[torch.nn.Parameter] = torch.nn.Parameter) -> Tuple[torch.Tensor, torch.Tensor]:
    """Binary weight initialisation (reference :286-327): scale_w = mean|w|; the centred weight goes through nv_tensor_quant
    with its default amax -- the SIGNED maximum of the centred weight, torch.amax, quant_operators.py:52 -- into int8 sign
    carriers, and exact zeros take sign(w)."""
    from bitorch_engine.utils.quant_operators import nv_tensor_quant
    w = weight.detach()
    if w.dtype != torch.float:
        w = w.to(torch.float)
    scale_w = w.norm(p=1).div(w.nelement()).to(weight.device)
    centred = w - w.mean()
    carriers = nv_tensor_quant(centred)[0]
    carriers = torch.where(carriers == 0, centred.sign(), carriers)
    return cls(carriers.to(torch.int8), requires_grad=False), scale_w


def _forget_conversions(t: torch.Tensor) -> None:
    """Drop what the extension shims memoised ON the tensor (packed rows, FP4 / tap / lane images: extensions.q_linear_cuda._cached keys on
    (version, address), and an optimiser-side write through `.data` moves neither).  Called after every update of an integer parameter."""
    try:
        t.__dict__.pop("_bie_memo", None)
    except AttributeError:
        pass


def _unpack_gptq_zeros(qzeros: torch.Tensor, w_bit: int, n_cols: int) -> torch.Tensor:
    """int32 [G, N*w/32] packed along N -> integer zeros + 1, [G, N] (the zeros half of gptq_style_unpacking,
    reference utils/quant_operators.py:326-331)."""
    wf = torch.arange(0, 32, w_bit, dtype=torch.int32, device=qzeros.device)
    z = torch.bitwise_right_shift(qzeros.unsqueeze(2).expand(-1, -1, 32 // w_bit), wf.view(1, 1, -1))
    z = torch.bitwise_and(z, (2 ** w_bit) - 1) + 1
    return z.reshape(-1, n_cols)


def _pack_gptq_zeros(zeros: torch.Tensor, w_bit: int, out_features: int) -> torch.Tensor:
    """gptq_style_zeros_packing (reference utils/quant_operators.py:348-368): truncation to int32, stored value - 1."""
    z = zeros.reshape(zeros.shape[0], math.ceil(out_features // 32 * w_bit), 32 // w_bit).to(torch.int32)
    wf = torch.arange(0, 32, w_bit, device=zeros.device, dtype=torch.int32)
    z = torch.bitwise_and(z - 1, (2 ** w_bit) - 1)
    return torch.bitwise_left_shift(z, wf.view(1, 1, -1)).sum(dim=-1).to(torch.int32)


def _add_scaled_(t: torch.Tensor, other: torch.Tensor, alpha: float) -> torch.Tensor:
    """`t.add_(other, alpha=alpha)` with the rounding of torch's CPU kernel, which is what the reference's update runs on and what the
    golden vectors pin: for fp16 / bf16 tensors the scalar `alpha` is first rounded to the TENSOR dtype, then t + alpha * other is
    evaluated in fp32 (the product of two 16-bit floats is exact there, so fused or not does not matter) and rounded once.  The GPU
    kernel of the same torch op keeps alpha in fp32 -- a third of the first-moment elements then differ in the last bit."""
    if t.dtype in (torch.float16, torch.bfloat16):
        a = float(torch.tensor(alpha, dtype=t.dtype))
        return t.copy_((t.float() + a * other.float()).to(t.dtype))
    return t.add_(other, alpha=alpha)


def _update_integer_parameter(qweight, exp_avg_s, exp_avg_l, step, lr, weight_decay, beta1, beta2, eps, dtype, correct_bias) -> None:
    """The branches of the reference's qweight_update_fn for parameters whose gradient is an INTEGER (or boolean) tensor in `qweight.grad`
    (utils/model_helper.py:403-478).  Plain torch elementwise ops in the reference, and here: the same ops in the same order, so the
    roundings land where the reference's do (pinned to reference outputs: tests/golden/update_step_integer_params.npz).  Stock torch's
    autograd cannot PRODUCE such a gradient (the reference runs a patched torch); whoever computes it assigns `.grad` (with
    `grad_dtype = None` where the data has been re-typed).
      * binary linear / conv (:436-444): first moment lerp-ed towards the gradient, second towards lr * sign(first); a weight keeps its sign
        where it equals -sign(second) (zero counts as +1) and flips elsewhere;
      * W4A4 / W8A8 (:450-478): Adam on the integer values in `dtype`, decoupled weight decay, then nv_tensor_quant -- the data comes back
        in `dtype` holding integers, as the reference leaves it;
      * boolean binary embedding table (:412-434): the second moment lerp-ed towards +-lr (gradient bits), its sign bits XOR-ed into the
        active rows.  The packed uint8 table cannot run in the reference either (:408 builds a shape from a tensor row)."""
    from bitorch_engine.layers.qembedding.binary.layer import BinaryEmbeddingParameter
    from bitorch_engine.layers.qlinear.binary.layer import BinaryLinearParameter
    from bitorch_engine.layers.qconv.binary.layer import BinaryConvParameter
    from bitorch_engine.layers.qlinear.nbit.layer import nBitLinearParameter
    from bitorch_engine.layers.qconv.nbit.layer import nBitConvParameter
    from bitorch_engine.utils.quant_operators import nv_tensor_quant
    if qweight.grad is None:
        raise RuntimeError("qweight_update_fn: qweight.grad is not set (integer gradients are assigned by the caller; stock autograd cannot produce them)")
    if isinstance(qweight, BinaryEmbeddingParameter):
        if qweight.data.dtype is not torch.bool:
            raise NotImplementedError("qweight.dtype '{}' has not been supported yet.".format(str(qweight.data.dtype)))
        towards = qweight.grad.to(dtype)
        towards = torch.where(towards == 0, torch.tensor(-1, dtype=dtype, device=qweight.device), towards).mul_(lr)
        exp_avg_s.lerp_(towards, (1 - beta2))
        bits = exp_avg_s >= 0
        rows = qweight.active_indices
        qweight[rows] ^= qweight[rows] ^ bits[rows]
        _forget_conversions(qweight)
        return
    if isinstance(qweight, (BinaryLinearParameter, BinaryConvParameter)):
        exp_avg_l.lerp_(qweight.grad.to(dtype), (1 - beta1))
        exp_avg_s.lerp_(exp_avg_l.clone().sign_().mul_(lr), (1 - beta2))
        keep = exp_avg_s.clone().sign_().mul_(-1)
        keep[keep == 0] = 1
        flip = keep != qweight.sign()
        qweight.data.copy_(torch.where(flip, -qweight.data, qweight.data))
        _forget_conversions(qweight)  # written through .data: neither the version counter nor the address moved
        return
    if isinstance(qweight, (nBitLinearParameter, nBitConvParameter)):
        g = qweight.grad.to(dtype)
        w = qweight.data.to(dtype)
        exp_avg_l.mul_(beta1).add_(g, alpha=(1.0 - beta1))
        exp_avg_s.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = exp_avg_s.sqrt().add_(eps)
        step_size = lr
        if correct_bias:
            step_size = step_size * math.sqrt(1.0 - beta2 ** step.item()) / (1.0 - beta1 ** step.item())
        w.addcdiv_(exp_avg_l, denom, value=-step_size)
        if weight_decay > 0.0:
            w.add_(w, alpha=(-lr * weight_decay))
        qweight.data = nv_tensor_quant(w)[0]
        _forget_conversions(qweight)
        return
    raise NotImplementedError("qweight.dtype '{}' has not been supported yet.".format(str(qweight.data.dtype)))


def qweight_update_fn(qweight: torch.nn.Parameter, exp_avg_s: torch.Tensor = None, exp_avg_l: torch.Tensor = None, step: torch.Tensor = None,
                      lr: float = 1e-4, weight_decay: float = 0.0, beta1: float = 0.99, beta2: float = 0.9999, eps: float = 1e-6,
                      dtype=torch.half, correct_bias=None, projector=None, grad: torch.Tensor = None) -> None:
    """The DiodeMix re-pack step of a quantised parameter, ON THE DEVICE: unpack -> Adam moments -> update -> (every fifth step)
    zero-point update -> pack.  Mirror of the reference's qweight_update_fn (utils/model_helper.py:363-532) for
    MPQWeightParameter in the GPTQ form (asym, explicit g_idx, layer_type 1) -- the case the reference executes: its symmetric
    g_idx branch returns an unassigned `zeros` (quant_operators.py:341-343) and the no-g_idx branch needs an MBWQ q_perm.
    The unpack is the HIP dequant kernel (bie_mpq_dequant == gptq_style_unpacking here: s * (q - (zq + 1)), one rounding), the pack
    the HIP pack kernel (bie_mpq_pack == pack_fp_weight: round(w / s + z), clamp, bit-pack); the moment arithmetic in between is
    the reference's own sequence of torch elementwise ops in `dtype`, op for op, so every rounding lands where the reference's does
    (pinned to reference outputs: tests/golden/update_step.npz).  Like the reference, pack_fp_weight sees the zero points from
    BEFORE this step's update_zeros.
    PARITY MODE: the golden vectors come from the reference run on the CPU (the only place it runs in this container), so two ops are spelled
    with torch's CPU-kernel roundings (_add_scaled_, the addcmul below).  The reference's MPQ update executes on CUDA tensors in practice,
    where torch keeps `alpha` in fp32 and fuses differently: about a third of the first-moment elements then differ in the last bit from this
    function, which can flip a re-packed integer now and then.  This is CPU-kernel parity, stated as such; GPU-kernel parity
    would need vectors generated by the reference on a GPU."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.layer import MPQWeightParameter
    step.add_(1)  # FIRST, as the reference (utils/model_helper.py:401: the counter moves before the parameter kind is looked at, also when a branch then raises)
    if not isinstance(qweight, MPQWeightParameter):
        return _update_integer_parameter(qweight, exp_avg_s, exp_avg_l, step, lr, weight_decay, beta1, beta2, eps, dtype, correct_bias)
    if not (qweight.layer_type == 1 and qweight.asym and qweight.g_idx is not None):
        raise NotImplementedError("qweight_update_fn: MPQWeightParameter is updated in its GPTQ form (layer_type 1, asym, g_idx), the form the "
                                  "reference's own update executes")
    w_bit, gs = qweight.w_bit, qweight.group_size
    scales, qzeros, g_idx = qweight.scales, qweight.zeros, qweight.g_idx
    K, N = g_idx.numel(), qweight.shape[1]
    w = q_linear_cuda.mpq_dequant(qweight.data, scales, qzeros, g_idx, w_bit, True, gs).to(dtype)  # == gptq_style_unpacking(qweight)[0]
    _add_scaled_(exp_avg_l.mul_(beta1), grad, 1.0 - beta1)
    # t.addcmul_(g, g, value) as torch's CPU kernel evaluates it: t + (value * g) * g, left to right in fp32 (every product and the
    # sum rounded to fp32, `value` kept in fp32), then one rounding to dtype.  Separate eager ops: no fused multiply-add can sneak
    # in.  (The GPU kernel of the op differs for bf16: the reference vectors caught it.)
    exp_avg_s.mul_(beta2)
    g32 = grad.float()
    exp_avg_s.copy_((exp_avg_s.float() + ((1.0 - beta2) * g32) * g32).to(exp_avg_s.dtype))
    denom = exp_avg_s.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        bias_correction1 = 1.0 - beta1 ** step.item()
        bias_correction2 = 1.0 - beta2 ** step.item()
        step_size = step_size * math.sqrt(bias_correction2) / bias_correction1
    norm_grad = exp_avg_l / denom
    if projector is not None:
        norm_grad = projector.project_back(norm_grad.to(dtype))
    _add_scaled_(w, norm_grad, -step_size)
    new_zeros = None
    if step % 5 == 0:  # update_zeros, layer_type 1 with g_idx (reference model_helper.py:346-356)
        gl = g_idx.long()
        zeros_unpack = _unpack_gptq_zeros(qzeros, w_bit, N).to(dtype)[gl]
        zeros_unpack.add_(step_size * norm_grad)
        perm = torch.argsort(gl, dim=0)
        zeros = zeros_unpack[perm, :].view(-1, K // scales.size(0), scales.size(-1)).mean(1)
        new_zeros = _pack_gptq_zeros(zeros, w_bit, zeros.size(-1))
    qweight.data = q_linear_cuda.mpq_pack(w, scales, qzeros, g_idx, w_bit, True, gs)  # the zero points from before update_zeros, as the reference
    if new_zeros is not None:
        qweight.zeros = new_zeros
