#!/usr/bin/env python3
"""HIP engine with precision "f32" vs the oracle's PM-F32 back-end with aten_all, picture by picture (spatial coder
only): symbols, CDF rows, bytes and the reconstruction must be identical.  GPU tool.  usage: [WxH] [q]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("learned-pmctf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["PMCTF_PRECISION"] = "f32-chain" if "--chain" in sys.argv else "f32"
import numpy as np, torch
from helpers import frames, product_model
from pmctf_oracle.model import Oracle, get_curr_q
from pMCTF.utils.stream_helper import image_header

size = next((a for a in sys.argv[1:] if "x" in a), "384x256")
W, H = (int(v) for v in size.split("x"))
q = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
net, sd = product_model(1)
eng = net.engine()
eng.keep_streams = True
orc = Oracle(sd, 1, "cdef", aten_all=eng.aten_all)
(ry, rc), (cy, cc) = frames(W, H, 2)
bad = 0
with torch.no_grad():
    for coder, name, x in (("hp_coder", "H luma", cy - ry), ("hp_coder", "H chroma", cc - rc), ("lp_coder", "L luma", cy),
                           ("lp_coder", "L chroma", cc)):
        N, _, h, w = x.shape
        x_hat, stream = eng.pwave_compress(coder, x.cuda(), q, None)
        ox, odata, otrace = orc.pwave_compress(coder, x, [1, N, h, w], q, None)
        size_, data, (sym, idx) = eng.coder.submit(stream, eng.tables, lambda n: image_header(h, w, N, n), None, True).result()
        osym = np.concatenate([t[0] for t in otrace]).ravel(); oidx = np.concatenate([t[1] for t in otrace]).ravel()
        ds, di = int((np.asarray(sym).ravel() != osym).sum()), int((np.asarray(idx).ravel() != oidx).sum())
        dx = int((x_hat.cpu().numpy().view(np.int32) != ox.numpy().view(np.int32)).sum())
        if di:
            first = int(np.nonzero(np.asarray(idx).ravel() != oidx)[0][0])
            lens = np.cumsum([t[0].size for t in otrace])
            print("   first differing row at element", first, "push", int(np.searchsorted(lens, first, side="right")), "of", len(otrace))
        print(f"{name:9s} {tuple(x.shape)}: symbols differing {ds}, CDF rows differing {di}, bytes identical {data == bytes(odata)}, "
              f"reconstruction elements differing {dx}", flush=True)
        bad += ds + di + dx
sys.exit(1 if bad else 0)
