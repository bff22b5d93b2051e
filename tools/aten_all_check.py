#!/usr/bin/env python3
"""The spatial coder of ONE picture through the oracle's two back-ends: ATen (what the reference computes) and PM-F32 with
ATen's summation order in every layer (aten_all = the product's precision "f32").  The two must write the same
bytes.  CPU only (test infrastructure).   usage: aten_all_check.py [WxH] [luma|chroma] [q_index] [H|L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from helpers import frames, synth_sd_cpu
from pmctf_oracle.model import Oracle

size = next((a for a in sys.argv[1:] if "x" in a), "1920x1080")
W, H = (int(v) for v in size.split("x"))
kind = "chroma" if "chroma" in sys.argv else "luma"
q = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
coder = "lp_coder" if "L" in sys.argv[1:] else "hp_coder"
torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
sd = synth_sd_cpu(1)
fr = frames(W, H, 2)
(ry, rc), (cy, cc) = fr
x = ((cy - ry) if kind == "luma" else (cc - rc)) if coder == "hp_coder" else (cy if kind == "luma" else cc)
x = x * (1.0 if coder == "hp_coder" else 1.0)
side = (0, x.size(0), x.size(2), x.size(3))
res = {}
with torch.no_grad():
    for name, o in (("aten", Oracle(sd, 1, "torch")), ("pm-f32 aten_all", Oracle(sd, 1, "cdef", aten_all=True)),
                    ("pm-f32", Oracle(sd, 1, "cdef"))):
        t0 = time.time()
        xh, data, trace = o.pwave_compress(coder, x, side, q)
        res[name] = (xh.numpy(), bytes(data), trace)
        print(f"{name:18s} {len(data)} bytes  {time.time() - t0:.0f} s", flush=True)
ref = res["aten"]
for name in ("pm-f32 aten_all", "pm-f32"):
    xh, data, trace = res[name]
    npush = len(ref[2])
    bad_sym = sum(int((np.asarray(a[0]) != np.asarray(b[0])).sum()) for a, b in zip(ref[2], trace))
    bad_idx = sum(int((np.asarray(a[1]) != np.asarray(b[1])).sum()) for a, b in zip(ref[2], trace))
    print(f"{name:18s} vs aten: bytes identical {data == ref[1]}  pushes {npush}  symbols differing {bad_sym}  "
          f"CDF rows differing {bad_idx}  reconstruction elements differing "
          f"{int((xh.view(np.int32) != ref[0].view(np.int32)).sum())} / {xh.size}", flush=True)
