#!/usr/bin/env python3
"""Micro-benchmark of the bandwidth-bound kernels on the shapes of the 1080p encode path: algorithmic bytes per launch
(the tensors the op must read and write once) over the HIP-event time -> GB/s against the 8 TB/s HBM3E peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch  # noqa: E402
from pMCTF.hip import ops  # noqa: E402

PEAK = 8000.0  # GB/s


def timed(fn, reps=20):
    """GPU time per launch: the repetitions are captured into one HIP graph and the replay is bracketed by events, so that
    a launch of a few microseconds is not measured at the rate the host can enqueue it"""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


ROWS = []      # (name, seconds per launch, algorithmic bytes): bench.py's `hbm_kernels` block reads these


def report(name, nbytes, t):
    ROWS.append((name, t, nbytes))
    print(f"{name:58s} {t * 1e6:8.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / t / 1e9:7.0f} GB/s  ({nbytes / t / 1e9 / PEAK * 100:4.1f} % of HBM peak)")


def main():
    dev = "cuda"
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)      # 1 GiB
    dst = torch.empty_like(big)
    report("reference point: device-to-device copy of 1 GiB (runtime's copy kernel)", 2 * big.numel() * 4,
           timed(lambda: dst.copy_(big), 5))
    del big, dst
    H, W = 1152, 1920
    f4 = 4
    y = torch.randn(1, 1, H, W, device=dev)
    flow = torch.randn(1, 2, H, W, device=dev) * 3
    lx, ly = torch.linspace(-1, 1, W, device=dev), torch.linspace(-1, 1, H, device=dev)
    y2 = torch.empty_like(y)
    report("reference point: copy of ONE 1152x1920 f32 plane (8 B/px; what a launch of this size can reach)", H * W * 8,
           timed(lambda: y2.copy_(y)))
    report("flow_warp 1x1x1152x1920 (4 B in, 8 B flow, 4 B out /px)", H * W * 16, timed(lambda: ops.flow_warp(y, flow, lx, ly)))
    a, b = torch.randn(1, 1, H, W, device=dev), torch.randn(1, 1, H, W, device=dev)
    report("ew add 1x1x1152x1920 (12 B/elem)", H * W * 12, timed(lambda: ops.ew(ops.EW_ADD, a, b)))
    report("even-row split of a 1152x1920 plane (strided view -> dense, 8 B/output elem)", H // 2 * W * 8,
           timed(lambda: ops.ew(ops.EW_COPY, a[:, :, ::2, :])))
    half = torch.randn(1, 1, H // 2, W, device=dev)
    report("transpose of a 576x1920 plane (8 B/elem)", H // 2 * W * 8,
           timed(lambda: ops.ew(ops.EW_COPY, half.permute(0, 1, 3, 2))))
    t112 = torch.randn(1, 576, 960, 112, device=dev)
    report("parity-class gather of a 1x576x960x112 tensor (8 B/output elem)", 288 * 480 * 112 * 8,
           timed(lambda: ops.ew(ops.EW_COPY, ops.as_nchw(t112)[:, :, 1::2, 0::2], out=ops.as_nchw(ops.empty_nhwc(1, 288, 480, 112, dev)))))
    del t112
    x112 = torch.randn(1, 576, 960, 112, device=dev)
    dw = ops.DepthwiseConv2d(torch.randn(112, 1, 3, 3), torch.randn(112))
    report("depthwise 3x3 1x576x960x112 (8 B/elem)", x112.numel() * 8, timed(lambda: dw(x112)))
    sub = torch.randn(1, 576, 960, 1, device=dev)
    c1 = ops.Conv2d(torch.randn(112, 1, 3, 3) * 0.1, torch.randn(112), 1, (1, 1))
    report("conv 1->112 3x3 1x576x960 (strip kernel; 4 B in + 448 B out /px)", 576 * 960 * 452, timed(lambda: c1(sub)))
    x16 = torch.randn(1, H, W, 16, device=dev)
    c16 = ops.Conv2d(torch.randn(16, 16, 3, 3) * 0.1, torch.randn(16), 1, (1, 1))
    report("conv 16->16 3x3 1152x1920 (persistent MFMA; 128 B/px)", H * W * 128, timed(lambda: c16(x16)))
    c161 = ops.Conv2d(torch.randn(1, 16, 3, 3) * 0.1, torch.randn(1), 1, (1, 1))
    report("conv 16->1 3x3 1152x1920 (few-cout kernel; 68 B/px)", H * W * 68, timed(lambda: c161(x16)))
    yin = torch.randn(1, H, W, 1, device=dev)
    c116 = ops.Conv2d(torch.randn(16, 1, 3, 3) * 0.1, torch.randn(16), 1, (1, 1))
    report("conv 1->16 3x3 + tanh, dual output 1152x1920 (132 B/px)", H * W * 132,
           timed(lambda: ops.conv3x3_cin1_dual(c116, yin, ops.ACT_TANH)))
    x64 = torch.randn(1, 576, 960, 64, device=dev)
    report("nearest x2 upsampling 1x576x960x64 (20 B/input elem)", x64.numel() * 20, timed(lambda: ops.nearest_up2(x64)))
    y3 = torch.randn(1, 3, H, W, device=dev)
    report("avg-pool 2x2 1x3x1152x1920 (5 B/input elem)", 3 * H * W * 5, timed(lambda: ops.avgpool2(y3)))
    return ROWS


if __name__ == "__main__":
    main()
