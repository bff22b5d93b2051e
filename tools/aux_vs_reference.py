#!/usr/bin/env python3
"""The auxiliary reduced-precision profiles (and the exact one) against the digests of the REAL reference's CPU runs at
every full-size BASELINE point: per rate point the frames whose bit count differs, the largest per-frame bit delta, the
relative change of the total and the largest PSNR error.  No parity claim is made for the auxiliary profiles; this is the
report that goes with them.  usage: aux_vs_reference.py [out.json]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import numpy as np, torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
W, H = 1920, 1080
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
points = [(8, 3), (16, 0), (16, 3), (16, 4), (16, 8), (16, 12), (16, 16), (16, 20)]
out = {}
tmp = tempfile.mkdtemp()
frames = {g: [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, g)] for g in (8, 16)}
for prec in ("f32", "bf16x3", "bf16x2", "bf16"):
    net.precision = prec
    for gop, q in points:
        fix = os.path.join(ROOT, "tests", "golden", "reference_1920x1080_gop%d_me4%s_digest.npz" % (gop, "" if q == 3 else f"_q{q}"))
        g = np.load(fix)
        with torch.no_grad():
            t0 = time.time()
            enc = pmctf_gop.encode_gop_batched(net, frames[gop], H, W, q, tmp)
            torch.cuda.synchronize(); dt = time.time() - t0
            ps = pmctf_gop.gop_psnr(pmctf_gop.decode_gop(net, [list(f) for f in enc["frames_coded"]]), frames[gop], H, W)
        db = np.array(enc["bits"]) - g["gop.bits"]
        out.setdefault(prec, {})[f"gop{gop}_q{q}"] = {
            "frames_with_bit_delta": int((db != 0).sum()), "frames": int(db.size), "max_abs_bit_delta": float(np.abs(db).max()),
            "rel_total_bits": float(db.sum() / g["gop.bits"].sum()),
            "max_abs_dpsnr_db": float(np.abs(np.array([p["yuv"] for p in ps]) - g["gop.psnr_yuv"]).max()),
            "bpp": float(sum(enc["bits"]) / (gop * W * H)), "bpp_reference": float(g["gop.bits"].sum() / (gop * W * H)),
            "encode_s_stage_batched_incl_first_use": dt}
        r = out[prec][f"gop{gop}_q{q}"]
        print(f"{prec:7s} gop{gop:2d} q{q:2d}: {r['frames_with_bit_delta']:2d}/{r['frames']} frames differ, max |dbits| {r['max_abs_bit_delta']:9.0f}, "
              f"rel total {r['rel_total_bits']:+.2e}, max |dPSNR| {r['max_abs_dpsnr_db']:.2e} dB", flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
