#!/usr/bin/env python3
"""1080p decode timing for one pair (skip_decoding=False)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
from pMCTF.hip import engine as E
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
fr = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, 2)]
tmp = tempfile.mkdtemp()
with torch.no_grad():
    for it in range(2):
        dpb = {"mv_feature": None, "ref_mv_y": None}
        r = net.encode_one_stage(fr[0], fr[1], True, dpb, output_path=os.path.join(tmp, "1.bin"), pic_width=W, pic_height=H,
                                 skip_decoding=False, stage_idx=0, q_index=3)
        print("decoding_time", r["decoding_time"], flush=True)
