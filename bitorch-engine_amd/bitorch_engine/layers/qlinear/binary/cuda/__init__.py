from .bmm import BMM
from .layer import BinaryLinearCuda
