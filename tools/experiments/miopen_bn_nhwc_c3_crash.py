"""Reproducer (GPU box): torch.nn.BatchNorm2d(3) in training mode on a bf16 channels-last input segfaults the host inside
MIOpen at batch sizes below 4 (ROCm 7.2 image, torch 2.10; also 64 channels at 27 x 27); NCHW input or batch >= 4 are fine.
    python tools/experiments/miopen_bn_nhwc_c3_crash.py 2 3 254        -> Segmentation fault
    python tools/experiments/miopen_bn_nhwc_c3_crash.py 4 3 254        -> ok
    python tools/experiments/miopen_bn_nhwc_c3_crash.py 2 3 254 nchw   -> ok
    python tools/experiments/miopen_bn_nhwc_c3_crash.py 3 64 27        -> Segmentation fault
Image_Encoder._block_nhwc therefore runs training-mode BatchNorm on NCHW copies when the batch is smaller than 4."""
import faulthandler
import sys

import torch

faulthandler.enable()
B, C, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
bn = torch.nn.BatchNorm2d(C).to(dev)
x = torch.rand(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
if len(sys.argv) > 4:
    x = x.contiguous()
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = bn(x)
torch.cuda.synchronize()
print("bn", B, C, H, "ok", y.dtype, y.stride())
