"""Binary conv 512->512 3x3 on 7x7 (BASELINE.json configs[3]) and the other ResNet stage shapes: the one-launch form (binary_conv_fused.hip) against the
forms it replaces (BIE_CONV_FUSED_MAX_ROWS=0: tap kernels / FP4 GEMM), us per call, 16 calls per graph replay."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import _binary_common as bc
dev = torch.device("cuda", 0)
out = []
shapes = [(1, 512, 7, 512, 3, 1, 1), (8, 512, 7, 512, 3, 1, 1), (32, 512, 7, 512, 3, 1, 1), (128, 512, 7, 512, 3, 1, 1), (512, 512, 7, 512, 3, 1, 1),
          (32, 256, 14, 256, 3, 1, 1), (32, 128, 28, 128, 3, 1, 1), (32, 256, 14, 512, 3, 2, 1), (32, 256, 14, 512, 1, 2, 0)]
for (B, C, H, OC, ks, st, pad) in shapes:
    x = torch.randn((B, C, H, H), device=dev)
    w = torch.randn((OC, C, ks, ks), device=dev)
    wp = bc.pack_rows(w.reshape(OC, -1)).contiguous()
    row = {"B": B, "C": C, "H": H, "OC": OC, "k": ks, "stride": st}
    for name, fused, mfma in (("one_launch_valu", "100000000", "0"), ("one_launch_mfma", "0", "100000000"), ("replaced_forms", "0", "0")):
        os.environ["BIE_CONV_FUSED_MAX_ROWS"], os.environ["BIE_CONV_MFMA_MAX_ROWS"] = fused, mfma
        fn = lambda s_: [bc.conv2d(x, wp, OC, ks, st, pad, 1, 1.0) for _ in range(16)]
        row[name] = round(time_graph(capture(fn), 10) / 16, 2)
    oh = (H + 2 * pad - ks) // st + 1
    row["GOP"] = round(2.0 * B * oh * oh * OC * C * ks * ks / 1e9, 2)
    row["valu_floor_us"] = round(row["GOP"] * 1e9 / 1260e12 * 1e6, 2)
    out.append(row)
    print(json.dumps(row), flush=True)
