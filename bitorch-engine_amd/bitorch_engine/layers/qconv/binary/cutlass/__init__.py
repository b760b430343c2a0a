from .layer import BinaryConv2dCutlass
