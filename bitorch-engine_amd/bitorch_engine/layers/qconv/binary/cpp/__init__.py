from .layer import BinaryConv2dCPP
