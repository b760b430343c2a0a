"""Model surgery helpers, API of reference utils/convert.py: find layers by type, swap them for quantised layers, and the
named MPQ configurations ("w_bit-group_size-dq_group_size")."""
from typing import Callable, Dict, Iterable, List, Optional, Type

import torch
from torch import nn

_MPQ_STRATEGIES = {  # w_bit, group_size, dq_group_size  (reference utils/convert.py:94-119)
    "2-8-32": (2, 8, 32),
    "2-32-32": (2, 32, 32),
    "2-128-32": (2, 128, 32),
    "4-128-256": (4, 128, 256),
    "8-128-256": (8, 128, 256),
}


def collect_layers(module: nn.Module, layer_types=None, parent_name: str = "") -> Dict[str, nn.Module]:
    """{dotted name: layer} of every sub-module whose exact type is in `layer_types` (default: nn.Linear) -- reference :7-31."""
    wanted = tuple(layer_types) if layer_types is not None else (nn.Linear,)
    if type(module) in wanted:
        return {parent_name: module}
    found: Dict[str, nn.Module] = {}
    for child_name, child in module.named_children():
        found.update(collect_layers(child, wanted, f"{parent_name}.{child_name}" if parent_name else child_name))
    return found


def replace_layers(module: nn.Module, names_to_replace: Iterable[str], class_: Type, replace_fn: Callable[[nn.Module], nn.Module],
                   parent_name: str = "") -> List[nn.Module]:
    """Replace, recursively, every attribute of `module` whose dotted name is in `names_to_replace` by replace_fn(old layer);
    returns the new layers -- reference :34-83."""
    if isinstance(module, class_):
        return []
    names = set(names_to_replace)
    done: List[nn.Module] = []
    for child_name, child in list(module.named_children()):
        full = f"{parent_name}.{child_name}" if parent_name else child_name
        if full in names:
            new = replace_fn(child)
            if not isinstance(new, class_):
                raise AssertionError("The replacement function does not create an object of the correct class. Recursion could fail/loop.")
            setattr(module, child_name, new)
            done.append(new)
        else:
            done.extend(replace_layers(child, names, class_, replace_fn, full))
    return done


def get_mpq_config(mpq_strategy: Optional[str] = None) -> dict:
    """Constructor kwargs of an MPQ layer for a strategy string (default "2-32-32") -- reference :94-119."""
    key = "2-32-32" if mpq_strategy is None else mpq_strategy
    assert key in _MPQ_STRATEGIES, f"{key} unknown!"
    w_bit, group_size, dq_group_size = _MPQ_STRATEGIES[key]
    return {"dq_mode": 2, "use_gba_quant": True, "asym": False, "w_bit": w_bit, "group_size": group_size, "dq_group_size": dq_group_size}


def quantize_linear_with_mpq_linear_cuda(module: nn.Module, names_to_replace: Iterable[str], mpq_strategy: Optional[str] = None,
                                         dtype: torch.dtype = torch.bfloat16, parent_name: str = ""):
    """nn.Linear -> MPQLinearCuda for the named layers -- reference :122-153."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    cfg = get_mpq_config(mpq_strategy)
    make = lambda old: MPQLinearCuda(in_channels=old.in_features, out_channels=old.out_features, dtype=dtype, **cfg)
    return replace_layers(module, names_to_replace, MPQLinearCuda, make, parent_name)


def _same_shape(class_):
    return lambda old: class_(old.in_features, old.out_features)


def quantize_linear_with_q4_linear_cutlass(module: nn.Module, names_to_replace: Iterable[str], parent_name: str = ""):
    """nn.Linear -> Q4LinearCutlass -- reference :156-175."""
    from bitorch_engine.layers.qlinear.nbit.cutlass import Q4LinearCutlass
    return replace_layers(module, names_to_replace, Q4LinearCutlass, _same_shape(Q4LinearCutlass), parent_name)


def quantize_linear_with_binary_linear_cuda(module: nn.Module, names_to_replace: Iterable[str], parent_name: str = ""):
    """nn.Linear -> BinaryLinearCuda -- reference :178-196."""
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    return replace_layers(module, names_to_replace, BinaryLinearCuda, _same_shape(BinaryLinearCuda), parent_name)
