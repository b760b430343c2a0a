from .functions import (fp32toint4, tensor_to_packed_uint8, unpack_uint8_tensor, q4_pack_tensor, q4_unpack_tensor,
                        q4_unpack_and_scaling_tensor)
