from .layer import BinaryLinearCPP
