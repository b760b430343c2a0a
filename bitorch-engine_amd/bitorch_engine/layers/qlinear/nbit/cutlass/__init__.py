from .q4_layer import Q4LinearCutlass, Q4MatMul
from .q8_layer import Q8LinearCutlass
