#!/usr/bin/env python3
"""The content-adaptive search (pmctf_ca.search_gop = test_pMCTF_CA.py:341-414) at full size on the GPU: 16 frames of
1920x1080, GOP sizes {16, 8, 4} x motion at 1, 1/2, 1/4, 1/8 resolution, in write mode and in estimate mode; prints the
options tried, their RD cost, the choice and the time per trial."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import numpy as np, torch
import pmctf_ca, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
W, H, G = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
# un-padded frames, as the harness reads them (test_pMCTF_CA.py:352-366)
fr = []
for y, u, v in pmctf_synth.synth_yuv420(W, H, G):
    Y = torch.from_numpy(y.astype(np.float32))[None, None].cuda()
    UV = torch.stack([torch.from_numpy(u.astype(np.float32)), torch.from_numpy(v.astype(np.float32))])[:, None].cuda()
    fr.append([Y, UV])
for write in (True, False):
    t_last = [time.time()]
    def on_trial(size, ds, logs):
        torch.cuda.synchronize(); now = time.time()
        print(f"  {'write' if write else 'estimate'} mode: GOP {size:2d}, motion 1/{ds}: bpp {float(np.mean(logs['bpps'])):.4f} "
              f"PSNR {float(np.mean(logs['psnrs'])):.3f} dB  rd {float(logs['rd']):.4f}  ({now - t_last[0]:.1f} s)", flush=True)
        t_last[0] = now
    with tempfile.TemporaryDirectory() as td, torch.no_grad():
        t0 = time.time()
        r = pmctf_ca.search_gop(net, fr, H, W, 3, td, write_stream=write, on_trial=on_trial)
        print(f"{'write' if write else 'estimate'} mode -> GOP {r['gop_choice']}, motion 1/{r['ds_choice']}, "
              f"{r['tested_opts']} options tried in {time.time() - t0:.1f} s")
