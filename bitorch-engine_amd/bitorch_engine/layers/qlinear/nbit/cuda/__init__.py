from .mpq_layer import MPQLinearCuda, MPQLinearCudaFunction
from .mbwq_layer import MBWQLinearCuda, MBWQLinearCudaFunction
from .utils import unpack_qweight, pack_fp_weight, make_group_map
from .mpq_list import MPQForwardList, MBWQExl2ForwardList
