#!/usr/bin/env python3
"""One warm-up GOP and N timed GOPs of the 1080p GOP-16 encode through the literal harness schedule (encode_one_stage
pair by pair, every bit count looked at when the call returns).  For rocprofv3 traces (tools/idle_gaps.py) and for
quick A/B runs of engine switches given as environment variables.  usage: eager_gop.py [gops] [lazy]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
from pMCTF.hip import lib as _lib
for kv in os.environ.get("CONV_OPTIONS", "").split(","):       # NAME=VALUE launch-shape knobs (pmctf_conv2d_set_option)
    if kv:
        k, v = kv.split("=")
        assert _lib.hip().pmctf_conv2d_set_option(k.encode(), int(v)) == 0
net.lazy_stages = len(sys.argv) > 2 and sys.argv[2] == "lazy"
W, H, G = 1920, 1080, 16
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, G)]
tmp = tempfile.mkdtemp()
look = lambda s, a, b, r: float(r["bit_H"] + r["bit_ME"])
with torch.no_grad():
    pmctf_gop.encode_gop(net, frames, H, W, 3, tmp, on_pair=look); torch.cuda.synchronize()
    eng = net.engine()
    for k in eng.stats: eng.stats[k] = 0
    if os.environ.get("PLAN_TIMING"):
        eng.plan_timing = []
    t = time.time()
    for _ in range(n):
        enc = pmctf_gop.encode_gop(net, frames, H, W, 3, tmp, on_pair=look)
    torch.cuda.synchronize(); t = time.time() - t
if eng.plan_timing:
    print("per pair, ms after the call's first launch: motion codec done | luma analysis done | chroma analysis done | luma synthesis done | chroma synthesis done")
    for ev in eng.plan_timing[:15]:
        print("   " + "  ".join(f"{ev[0].elapsed_time(e):7.1f}" for e in ev[1:6]) + f"   motion est. {ev[6].elapsed_time(ev[0]):6.1f}")
    gaps = [a[7].elapsed_time(b[6]) for a, b in zip(eng.plan_timing, eng.plan_timing[1:])]
    print("GPU idle between the last kernel of a pair and the first of the next (ms):", " ".join(f"{g:.2f}" for g in gaps[:16]),
          f" total {sum(gaps):.1f} over {len(gaps)} gaps")
print(f"memory: allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, "
      f"peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; plans {len(eng.pair_plans)}")
print(f"{G * n / t:.3f} frames/s  ({t / n * 1e3:.1f} ms per GOP)  bits {sum(enc['bits']):.0f}  stats {eng.stats}")
