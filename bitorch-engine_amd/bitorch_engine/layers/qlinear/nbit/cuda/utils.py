"""Device-side twins of the reference helpers in layers/qlinear/nbit/cuda/utils.py: unpack_qweight
(:5-69), pack_fp_weight (:72-147) run as HIP kernels (bit-exact with the reference's torch code,
see tests/), make_group_map (:150-187) is host bookkeeping."""
import torch

from bitorch_engine.extensions import q_linear_cuda as _ext


def _attr(qweight, name):
    v = getattr(qweight, name, None)
    if v is None and name == "layer_type":
        raise ValueError("Error: invalid attribute of qweight in 'unpack_qweight'.")
    return v


def unpack_qweight(qweight) -> torch.Tensor:
    """Dense [K, N] weight in the layer dtype reconstructed from a packed MPQWeightParameter."""
    layer_type = _attr(qweight, "layer_type")
    if layer_type == 1:
        K = qweight.shape[0] * 32 // qweight.w_bit
        G = qweight.scales.shape[0]
        return _ext.mpq_dequant(qweight.data, qweight.scales, qweight.zeros, qweight.g_idx, qweight.w_bit,
                                qweight.asym, (K + G - 1) // G)
    if layer_type == 2:
        if qweight.q_group_map is None:
            return _ext.mbwq_q42fp_weight(qweight.data, qweight.scales, qweight.zeros, qweight.group_size,
                                          qweight.w_bit, qweight.q_perm)
        return _ext.mbwq_exl2fp_weight(qweight.data, qweight.scales, qweight.zeros, qweight.q_perm,
                                       qweight.q_group_map, qweight.rows)
    raise NotImplementedError("Error: 'layer_type' not yet supported!")


def pack_fp_weight(weight: torch.Tensor, qweight, unpacked_zeros: torch.Tensor = None) -> torch.Tensor:
    """Quantise + bit-pack a dense [K, N] weight with the scales / zeros attached to `qweight`."""
    layer_type = getattr(qweight, "layer_type", None)
    if layer_type is None:
        raise ValueError("Error: invalid 'layer_type' attribute in 'unpack_qweight' method.")
    if not (layer_type == 1 or (layer_type == 2 and qweight.q_group_map is None)):
        raise NotImplementedError("Error: pack_fp_weight for MBWQLinear using channel-mix quantization not supported yet.")
    zeros = qweight.zeros
    if qweight.asym:
        if unpacked_zeros is not None:
            from bitorch_engine.utils.quant_operators import gptq_style_zeros_packing
            zeros = gptq_style_zeros_packing(unpacked_zeros, qweight.w_bit, weight.shape[1], qweight.group_size)
        elif zeros.dtype != torch.int32:
            raise ValueError("Error: Got invalid dtype of qweight.zeros while packing fp weight.")
    K = weight.shape[0]
    G = qweight.scales.shape[0]
    q_perm = getattr(qweight, "q_perm", None)
    if not qweight.asym and qweight.g_idx is None and q_perm is not None:
        weight = weight[q_perm.long()]  # reference gathers rows by q_perm first (utils.py:126-128)
    return _ext.mpq_pack(weight.to(qweight.scales.dtype), qweight.scales, zeros, qweight.g_idx, qweight.w_bit,
                         qweight.asym, (K + G - 1) // G)


def make_group_map(q_groups: torch.Tensor, num_qrows: int) -> torch.Tensor:
    """int16 [2K]: for every unpacked row k, (group index, rows left in that group)."""
    qg = q_groups.detach().cpu().to(torch.int64)
    n = qg.numel() // 2
    bits, starts = qg[0::2], qg[1::2]
    ends = torch.cat([starts[1:], torch.tensor([num_qrows])])
    rows = (ends - starts) * 32 // bits
    idx = torch.repeat_interleave(torch.arange(n), rows)
    left = torch.cat([torch.arange(int(r), 0, -1) for r in rows.tolist()]) if n else torch.zeros(0, dtype=torch.int64)
    return torch.stack([idx, left], dim=1).reshape(-1).to(torch.short).to(q_groups.device)
