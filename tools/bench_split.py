#!/usr/bin/env python3
"""Reduced-precision conv profile (conv_split.hip) against the exact f32 kernel on the dominant shapes: time, equivalent
TFLOP/s (FLOPs of the convolution, not of the split products) and the deviation from the exact result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
from pMCTF.hip import ops
SHAPES = [("ctx112 8x576x960", 8, 576, 960, 112, 112), ("ctx112 1x576x960", 1, 576, 960, 112, 112),
          ("ctx112 8x288x480", 8, 288, 480, 112, 112), ("ctx112 8x144x240", 8, 144, 240, 112, 112),
          ("post64 8x1152x1920", 8, 1152, 1920, 64, 64), ("post64 1x1152x1920", 1, 1152, 1920, 64, 64)]
torch.manual_seed(0)
ops.SPLIT_MIN_PX = 0
for name, N, H, W, Cin, Cout in SHAPES:
    w = torch.randn(Cout, Cin, 3, 3) * 0.05
    b = torch.randn(Cout)
    x = torch.randn(N, H, W, Cin, device="cuda")
    ref = None
    for ns in (0, 3, 2, 1):
        conv = ops.Conv2d(w, b, 1, (1, 1), split=ns)
        y = conv(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv(x, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        fl = 2.0 * N * H * W * Cout * Cin * 9
        if ns == 0:
            ref = y.clone(); err = 0.0
        else:
            err = (y - ref).abs().max().item() / ref.abs().max().item()
        print(f"{name:22s} {'f32 exact' if ns == 0 else 'bf16 x%d' % ns:10s} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s-equivalent  max rel dev {err:.2e}")
