#!/usr/bin/env python3
"""Time the fused PredictUpdate launch (pu_fused.hip) against the chain of separate launches it replaces, on the plane
shapes of a 1080p encode.  HIP events on the launch stream; algorithmic work 2 x 16x16x9 MAC per pixel on the matrix
pipe, 8-12 B per pixel of HBM traffic."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import product_model
net, _ = product_model(1)
eng = net.engine()
wt = "hp_coder.wavelet_transform.lift_h"
shapes = [(1, 1152, 1920), (2, 576, 960), (1, 576, 1920), (1, 960, 576), (8, 576, 1920), (1, 288, 960), (1, 72, 240), (2, 36, 120)]
def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, h, w) in shapes:
    x = torch.rand(n, 1, h, w, device="cuda") * 255
    o = torch.randn(n, 1, h, w, device="cuda") * 30
    res = {}
    for fused in (True, False):
        eng.pu_fused = fused; eng.pu_fused_max_px = 1 << 40
        res[fused] = (run(lambda: eng.predict_filter(0, x)), run(lambda: eng.lift_step(wt, "conv_P1", "P_1", x, o, 1.0)))
    px = n * h * w
    gf = px * 2 * 16 * 16 * 9 * 2 / 1e9
    print(f"{n}x{h}x{w}: temporal fused {res[True][0]:8.1f} us ({gf / res[True][0] * 1e-3:6.1f} TF/s, {px * 8 / res[True][0] * 1e-6:5.2f} TB/s)  unfused {res[False][0]:8.1f} us |"
          f" lift fused {res[True][1]:8.1f} us  unfused {res[False][1]:8.1f} us")
