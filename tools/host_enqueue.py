#!/usr/bin/env python3
"""Host time spent ENQUEUEING each temporal stage of the default (deferred) drop-in path against the GOP's wall time:
is the Python side ever the bottleneck?"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_gop, pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
W, H, G = 1920, 1080, 16
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, G)]
tmp = tempfile.mkdtemp()
log = []
orig = net.encode_stage_pairs
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); log.append((len(a[0]), time.perf_counter() - t, k.get("wait_files", True))); return r
net.encode_stage_pairs = timed
with torch.no_grad():
    pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); torch.cuda.synchronize()
    for rep in range(2):
        log.clear()
        t = time.perf_counter(); pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); t1 = time.perf_counter() - t
        torch.cuda.synchronize(); t2 = time.perf_counter() - t
        print(f"GOP returned after {t1*1e3:.0f} ms, GPU drained after {t2*1e3:.0f} ms; stage enqueue times (pairs, ms, waited):",
              [(p, round(s * 1e3), w) for p, s, w in log], "sum", round(sum(s for _, s, _ in log) * 1e3), "ms")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.no_grad():
        pr.enable(); pmctf_gop.encode_gop(net, frames, H, W, 3, tmp); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
