from .layer import BinaryConv2dBase, BinaryConvParameter
