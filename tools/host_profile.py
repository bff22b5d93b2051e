#!/usr/bin/env python3
"""Where does a pair's wall time go on the host side? (enqueue / GPU completion / range-coder tail)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import numpy as np, torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
from pMCTF.hip import lib
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
W, H, G = 1920, 1080, 16
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, G)]
tmp = tempfile.mkdtemp()
with torch.no_grad():
    pmctf_gop.encode_gop(net, frames, H, W, 3, tmp)
    eng = net.engine(); eng.profile_host = True
    for k in eng.stats: eng.stats[k] = 0
    t = time.time(); enc = pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); torch.cuda.synchronize(); t = time.time() - t
print("GOP wall", t, eng.stats)
# raw coder throughput on the last luma stream
sizes = {f: os.path.getsize(os.path.join(tmp, f)) for f in sorted(os.listdir(tmp))[:6]}
print(sizes)
