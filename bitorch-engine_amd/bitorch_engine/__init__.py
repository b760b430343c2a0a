"""bitorch_engine -- MI355X-native drop-in for the low-bit Q-Linear / Q-Conv hot path of
GreenBitAI/bitorch-engine.  Same module paths, class names, constructor arguments, state_dict keys
and packed-weight layout as the reference's `bitorch_engine` package; the arithmetic runs in
hand-written HIP kernels for gfx950 reached through the C ABI in include/bie_hip.h.

Stock PyTorch cannot hold gradients on integer parameters (the reference ships a patched torch,
reference bitorch_engine/__init__.py:10-29), so the packed `qweight` parameters default to
requires_grad=False here: this build targets inference.
"""
__version__ = "0.1.0"
