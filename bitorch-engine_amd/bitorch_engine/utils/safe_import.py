"""import_extension(name): same entry point as the reference's utils/safe_import.py:75-112.  The
extension objects are thin ctypes front-ends; a missing/unbuilt libbie_hip.so raises on first USE
(like the reference's ExtensionModulePlaceholder), never silently falls back."""
import importlib

KNOWN = ("q_linear_cuda", "binary_linear_cpp", "binary_linear_cuda", "binary_linear_cutlass",
         "binary_conv_cpp", "binary_conv2d_cutlass", "functions_cuda", "q_linear_cutlass", "q4_conv_cutlass")


class ExtensionModulePlaceholder:
    def __init__(self, name, reason):
        self._name, self._reason = name, reason

    def __getattr__(self, item):
        raise RuntimeError(f"The extension '{self._name}' is not available in the MI355X build: {self._reason}")


def import_extension(module_name: str, not_yet_implemented: bool = False):
    if module_name not in KNOWN:
        return ExtensionModulePlaceholder(module_name, "no HIP implementation (out of scope of this build)")
    return importlib.import_module(f"bitorch_engine.extensions.{module_name}")
