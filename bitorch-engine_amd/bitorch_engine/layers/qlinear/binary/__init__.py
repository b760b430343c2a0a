from .layer import BinaryLinearBase, BinaryLinearParameter
