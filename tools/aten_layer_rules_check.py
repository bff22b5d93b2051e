#!/usr/bin/env python3
"""Every convolution of the spatial coder's entropy-parameter networks on the plane shapes of one frame size (luma and chroma,
all four levels): F.conv2d against the oracle's C convolution under the rule the policy predicts for it
(Oracle.sum_rule with aten_all) — optionally with another intra-op thread count for ATen.  CPU only (test infrastructure).
usage: aten_layer_rules_check.py WxH [threads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, torch.nn.functional as F
from helpers import synth_sd_cpu
from pmctf_oracle import clib
from pmctf_oracle.model import Oracle
W, H = (int(v) for v in sys.argv[1].split("x"))
pw, ph = (W + 127) // 128 * 128, (H + 127) // 128 * 128
sd = synth_sd_cpu(1)
o = Oracle(sd, 1, "cdef", aten_all=True)
torch.manual_seed(0)
if len(sys.argv) > 2: torch.set_num_threads(int(sys.argv[2])); print("threads", torch.get_num_threads())
seen = set()
for lvl in (3, 2, 1, 0):
    for n, div in ((1, 1), (2, 2)):
        h, w = ph // div >> (lvl + 1), pw // div >> (lvl + 1)
        for k, wt in o.sd.items():
            if not k.endswith(".weight") or wt.dim() != 4 or not k.startswith("lp_coder.") or "wavelet" in k or "dequant" in k:
                continue
            cout, cin, kh, kw = wt.shape
            groups = 1
            if ".depth_conv" in k and cin == 1:
                continue
            if f".context_fusion.{lvl}." not in k and "context_prediction" not in k:
                continue
            hh, ww = (2 * h, 2 * w) if "deconv" in k else (h, w)
            sig = (n, cin, cout, kh, hh, ww)
            if sig in seen: continue
            seen.add(sig)
            x = torch.randn(n, cin, hh, ww); b = o.sd.get(k[:-7] + ".bias")
            ref = F.conv2d(x, wt, b, padding=kh // 2).numpy()
            rule = o.sum_rule(k[:-7], x, wt, 1, 1)
            y = clib.conv2d(x.numpy(), wt.numpy(), None if b is None else b.numpy(), 1, (kh // 2, kh // 2), rule)
            m = float((y.view(np.int32) == ref.view(np.int32)).mean())
            if m < 1.0:
                alt = {r: round(float((clib.conv2d(x.numpy(), wt.numpy(), None if b is None else b.numpy(), 1, (kh // 2, kh // 2), r).view(np.int32) == ref.view(np.int32)).mean()), 3) for r in (0, 1, 2, 16, 32, 48, 64, 80, 96)}
                print("MISMATCH", k, sig, "rule", rule, round(m, 3), alt, flush=True)
print("checked", len(seen))
