#!/usr/bin/env python3
"""Per-signature census of the convolutions of one 1080p GOP-16 encode (stage-batched schedule): launches, time,
TFLOP/s (HIP events)."""
import collections, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
from pMCTF.hip import ops
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
W, H, G = 1920, 1080, 16
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, G)]
tmp = tempfile.mkdtemp()
sigs = []
def match(conv, x, stride):
    sigs.append((tuple(x.shape), conv.Cout, conv.KH, stride))
    return True
with torch.no_grad():
    pmctf_gop.encode_gop_batched(net, frames, H, W, 3, tmp); torch.cuda.synchronize()
    pmctf_gop.encode_gop_batched(net, frames, H, W, 3, tmp); torch.cuda.synchronize()
