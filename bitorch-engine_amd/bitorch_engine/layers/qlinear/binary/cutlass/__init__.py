from .layer import BinaryLinearCutlass, BinaryMatMul
