#!/usr/bin/env python3
"""Which elementwise launches does one 1080p GOP-16 encode issue? (op code, logical shape, count, call site)"""
import collections, os, sys, tempfile, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
from pMCTF.hip import ops, engine
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
net.engine().use_graphs = False          # count the launches themselves, not plan replays
W, H, G = 1920, 1080, 16
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, G)]
tmp = tempfile.mkdtemp()
cnt = collections.Counter()
orig = ops.ew
names = {v: k for k, v in vars(ops).items() if k.startswith("EW_")}
def rec(op, a, b=None, alpha=0.0, beta=0.0, out=None):
    fr = traceback.extract_stack(limit=4)
    site = " < ".join(f"{f.name}:{f.lineno}" for f in reversed(fr[:-1]))
    cnt[(names[op], tuple(a.shape), a.stride(1) == 1, site)] += 1
    return orig(op, a, b, alpha, beta, out)
with torch.no_grad():
    pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); torch.cuda.synchronize()
    ops.ew = rec; engine.ew = rec
    pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); torch.cuda.synchronize()
tot = sum(cnt.values())
print("ew launches per GOP:", tot)
def elems(k): 
    n = 1
    for d in k[1]: n *= d
    return n
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1] * elems(kv[0]))[:60]:
    print(f"{v:6d} x {k[0]:20s} {str(k[1]):24s} cl={k[2]!s:5s} {elems(k)*v/1e6:9.1f} Melem  {k[3]}")
