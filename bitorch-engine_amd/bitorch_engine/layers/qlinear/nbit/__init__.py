from .layer import MPQWeightParameter, MPQLinearBase, nBitLinearBase, nBitLinearParameter
