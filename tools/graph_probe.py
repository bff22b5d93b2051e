#!/usr/bin/env python3
"""Feasibility probe: one 1080p pair's spatial coders (luma + chroma, analysis + synthesis) as a HIP graph — replay
time against stream launches, one stream against two, bit-equality of the results."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
eng = net.engine()
W, H = 1920, 1080
fr = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(W, H, 2)]
(ry, rc), (cy, cc) = fr
dpb = {"mv_feature": None, "ref_mv_y": None}

def motion():
    return eng.compress_mv(ry, cy, dpb, stage_idx=0, q_index=3)

def coders(mv_hat, two):
    outs = {}
    def luma():
        r = eng.compress_one_stage(ry, cy, False, mv_hat, False, 0, 3, False, defer=True)
        outs["l"] = r
    def chroma():
        r = eng.compress_one_stage(rc, cc, False, mv_hat, True, 0, 3, False, defer=True)
        outs["c"] = r
    if two:
        main = torch.cuda.current_stream()
        s1 = eng.side_streams[0]
        s1.wait_stream(main)
        with torch.cuda.stream(s1):
            chroma(); outs["c"]["finish"]()
        luma(); outs["l"]["finish"]()
        main.wait_stream(s1)
    else:
        luma(); chroma(); outs["l"]["finish"](); outs["c"]["finish"]()
    return outs

def timeit(fn, n=5):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3

with torch.no_grad():
    mv = motion(); mv_hat = mv["mv_hat"]
    ref = coders(mv_hat, False); torch.cuda.synchronize()
    print("stream launches, one stream: %.2f ms" % timeit(lambda: coders(mv_hat, False)))
    print("stream launches, two streams: %.2f ms" % timeit(lambda: coders(mv_hat, True)))
    print("motion chain, stream launches: %.2f ms" % timeit(motion))
    for two in (False, True):
        g = torch.cuda.CUDAGraph()
        t0 = time.time()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = coders(mv_hat, two)
        print("capture took %.2f s" % (time.time() - t0))
        g.replay(); torch.cuda.synchronize()
        same = all(torch.equal(out[k][n], ref[k][n]) for k in "lc" for n in ("H_t_hat", "L_t", "H_t")) and \
            all(torch.equal(out[k]["H_stream"].sym, ref[k]["H_stream"].sym) and torch.equal(out[k]["H_stream"].idx, ref[k]["H_stream"].idx) for k in "lc")
        print(f"graph ({'two streams' if two else 'one stream'}): replay %.2f ms, identical to stream launches: {same}" % timeit(g.replay))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        m2 = motion()
    g.replay(); torch.cuda.synchronize()
    print("motion chain graph: replay %.2f ms, identical: %s" % (timeit(g.replay), torch.equal(m2["mv_hat"], mv["mv_hat"]) and torch.equal(m2["stream"].sym, mv["stream"].sym)))
    print("mem allocated %.1f GB reserved %.1f GB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
