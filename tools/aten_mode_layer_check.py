#!/usr/bin/env python3
"""Every convolution layer of the spatial coder (by parameter key) on random input: the HIP engine's launch with the rule
precision "f32" gives the layer vs the oracle's C convolution with the oracle's rule for it.  GPU tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["PMCTF_PRECISION"] = "f32"
import numpy as np, torch
from helpers import product_model
from pmctf_oracle import clib
from pmctf_oracle.model import Oracle

net, sd = product_model(1)
eng = net.engine()
orc = Oracle(sd, 1, "cdef", aten_all=True)
g = torch.Generator().manual_seed(1)
seen, bad = set(), 0
for k, w in orc.sd.items():
    if not k.endswith(".weight") or w.dim() != 4 or not (k.startswith("hp_coder.") or k.startswith("lp_coder.")):
        continue
    p = k[:-7]
    cout, cin, kh, kw = w.shape
    if kh != kw or (p + ".bias") not in orc.sd and False:
        continue
    for (n, h, wd) in ((1, 40, 56), (2, 24, 40), (1, 288, 480) if cin * cout <= 112 * 112 else (1, 72, 120)):
        sig = (cout, cin, kh, n, h, wd, p.split(".")[0])
        if sig in seen:
            continue
        seen.add(sig)
        groups = 1
        x = torch.randn(n, cin, h, wd, generator=g)
        if cin == 1 and orc.sd[k].shape[1] == 1 and ".depth" in p:
            continue
        try:
            with eng.reference_planes(n):
                y = eng.conv(p, 1, kh // 2)(x.permute(0, 2, 3, 1).contiguous().cuda())
        except Exception as e:            # depthwise / transposed layers are not built through eng.conv
            print("skip", p, tuple(w.shape), type(e).__name__, str(e)[:60]); continue
        rule = orc.sum_rule(p, x, w, 1, 1)
        b = orc.sd.get(p + ".bias")
        ref = clib.conv2d(x.numpy(), w.numpy(), None if b is None else b.numpy(), 1, (kh // 2, kh // 2), rule)
        d = int((y.permute(0, 3, 1, 2).cpu().numpy().view(np.int32) != ref.view(np.int32)).sum())
        if d:
            bad += 1
            print(f"DIFF {p:60s} w {tuple(w.shape)} x {(n, cin, h, wd)} rule {rule}: {d} / {ref.size}", flush=True)
print("layers/shapes checked", len(seen), "differing", bad)
