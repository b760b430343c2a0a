from .layer import nBitConv2dBase, nBitConvParameter
