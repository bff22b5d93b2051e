#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the shapes of the 1080p encode path (HIP events, TFLOP/s)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch  # noqa: E402
from pMCTF.hip import ops  # noqa: E402

SHAPES = [  # name, N, H, W, Cin, Cout, K, stride, pad
    ("ctx112 L0 luma 576x960", 1, 576, 960, 112, 112, 3, 1, 1),
    ("ctx112 L0 chroma 2x288x480", 2, 288, 480, 112, 112, 3, 1, 1),
    ("ctx112 L1 luma 288x480", 1, 288, 480, 112, 112, 3, 1, 1),
    ("ctx112 L2 luma 144x240", 1, 144, 240, 112, 112, 3, 1, 1),
    ("ctx112 L3 luma 72x120", 1, 72, 120, 112, 112, 3, 1, 1),
    ("ctx112 L1 chroma 2x144x240", 2, 144, 240, 112, 112, 3, 1, 1),
    ("ctx112 L2 chroma 2x72x120", 2, 72, 120, 112, 112, 3, 1, 1),
    ("ctx112 L3 chroma 2x36x60", 2, 36, 60, 112, 112, 3, 1, 1),
    ("batched L0 luma 8x576x960", 8, 576, 960, 112, 112, 3, 1, 1),
    ("batched 8x72x120", 8, 72, 120, 112, 112, 3, 1, 1),
    ("batched 4x72x120", 4, 72, 120, 112, 112, 3, 1, 1),
    ("batched 16x36x60", 16, 36, 60, 112, 112, 3, 1, 1),
    ("batched 8x36x60", 8, 36, 60, 112, 112, 3, 1, 1),
    ("batched 4x36x60", 4, 36, 60, 112, 112, 3, 1, 1),
    ("batched 2x36x60", 2, 36, 60, 112, 112, 3, 1, 1),
    ("batched 1x72x120", 1, 72, 120, 112, 112, 3, 1, 1),
    ("batched 4x144x240", 4, 144, 240, 112, 112, 3, 1, 1),
    ("batched 8x144x240", 8, 144, 240, 112, 112, 3, 1, 1),
    ("post64 chroma 2x576x960", 2, 576, 960, 64, 64, 3, 1, 1),
    ("lstm 32->32 2x288x480", 2, 288, 480, 32, 32, 3, 1, 1),
    ("pu 16->16 960x576", 1, 960, 576, 16, 16, 3, 1, 1),
    ("post64 luma 1152x1920", 1, 1152, 1920, 64, 64, 3, 1, 1),
    ("spynet 32->64 7x7 1152x1920", 1, 1152, 1920, 32, 64, 7, 1, 3),
    ("spynet 8->32 7x7 1152x1920", 1, 1152, 1920, 8, 32, 7, 1, 3),
    ("spynet 64->32 7x7 1152x1920", 1, 1152, 1920, 64, 32, 7, 1, 3),
    ("spynet 32->16 7x7 1152x1920", 1, 1152, 1920, 32, 16, 7, 1, 3),
    ("spynet 32->64 7x7 576x960", 1, 576, 960, 32, 64, 7, 1, 3),
    ("spynet 32->16 7x7 576x960", 1, 576, 960, 32, 16, 7, 1, 3),
] + [("spysmall %d->%d 7x7 %dx%d" % (ci, co, h, w), 1, h, w, ci, co, 7, 1, 3) for h, w in ((288, 480), (144, 240), (72, 120), (36, 60))
     for ci, co in ((32, 64), (64, 32), (32, 16))] + [
    ("pu 16->16 3x3 1152x1920", 1, 1152, 1920, 16, 16, 3, 1, 1),
    ("pu 16->1 3x3 1152x1920", 1, 1152, 1920, 16, 1, 3, 1, 1),
    ("1x1 112->112 576x960", 1, 576, 960, 112, 112, 1, 1, 0),
    ("1x1 112->112 8x576x960", 8, 576, 960, 112, 112, 1, 1, 0),
    ("1x1 112->112 8x144x240", 8, 144, 240, 112, 112, 1, 1, 0),
    ("lstm 32->32 576x960", 1, 576, 960, 32, 32, 3, 1, 1),
    ("lstm 32->32 8x576x960", 8, 576, 960, 32, 32, 3, 1, 1),
] + [("nscan %dx576x960" % n, n, 576, 960, 112, 112, 3, 1, 1) for n in (2, 3, 4, 6, 8, 12, 16)] \
  + [("nscan64 %dx1152x1920" % n, n, 1152, 1920, 64, 64, 3, 1, 1) for n in (2, 4, 8)] \
  + [("s2 112->112 3x3 %dx576x960" % n, n, 576, 960, 112, 112, 3, 2, 1) for n in (1, 4, 8)] \
  + [("s2 112->112 3x3 8x288x480", 8, 288, 480, 112, 112, 3, 2, 1)] \
  + [("s2small 112->112 %dx%dx%d" % (n, h, w), n, h, w, 112, 112, 3, 2, 1) for n, h, w in
     ((2, 288, 480), (8, 144, 240), (1, 288, 480), (4, 144, 240), (2, 144, 240), (8, 72, 120), (2, 72, 120))] \
  + [("few 16->2 7x7 1152x1920", 1, 1152, 1920, 16, 2, 7, 1, 3), ("few 16->2 7x7 576x960", 1, 576, 960, 16, 2, 7, 1, 3),
     ("few 16->1 3x3 8x1152x1920", 8, 1152, 1920, 16, 1, 3, 1, 1), ("few 64->1 3x3 8x1152x1920", 8, 1152, 1920, 64, 1, 3, 1, 1)] \
  + [("k1 %d->%d %dx%d" % (ci, co, h, w), 1, h, w, ci, co, 1, 1, 0) for ci, co, h, w in
     ((64, 256, 576, 960), (256, 64, 576, 960), (64, 64, 576, 960), (192, 768, 72, 120), (768, 192, 72, 120),
      (192, 192, 72, 120), (192, 192, 36, 60))]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else None
    torch.manual_seed(0)
    from pMCTF.hip import lib
    for kv in sys.argv[3:]:                     # NAME=VALUE launch-shape knobs (pmctf_conv2d_set_option)
        k, v = kv.split("=")
        assert lib.hip().pmctf_conv2d_set_option(k.encode(), int(v)) == 0
    print("options:", sys.argv[3:])
    for name, N, H, W, Cin, Cout, K, S, P in SHAPES:
        if only and only not in name:
            continue
        w = torch.randn(Cout, Cin, K, K) * 0.05
        b = torch.randn(Cout)
        conv = ops.Conv2d(w, b, S, (P, P), rule=int(os.environ.get("BENCH_RULE", "0")))   # summation rule (PMCTF_SUM_*)
        x = torch.randn(N, H, W, Cin, device="cuda")
        if os.environ.get("BENCH_RELU_INPUT"):     # activations as a ReLU leaves them (half zeros)
            x = torch.relu(x)
        y = conv(x)
        res = torch.randn_like(y) if os.environ.get("BENCH_RES") else None      # residual-add epilogue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            conv(x, res1=res, out=y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * N * y.shape[1] * y.shape[2] * Cout * Cin * K * K
        print(f"{name:34s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 157.3 * 100:5.1f} % of f32 MFMA peak)")


if __name__ == "__main__":
    main()
