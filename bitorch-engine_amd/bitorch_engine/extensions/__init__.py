"""Python objects named like the reference's compiled extension modules
(reference bitorch_engine/extensions/, loaded through utils/safe_import.import_extension):
q_linear_cuda, binary_linear_cpp, binary_linear_cuda, binary_linear_cutlass, binary_conv_cpp,
binary_conv2d_cutlass, functions_cuda.  Each exports the same functions with the same argument
meaning; they forward to libbie_hip.so via ctypes (bitorch_engine/_hip.py)."""
