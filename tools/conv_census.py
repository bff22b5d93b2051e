#!/usr/bin/env python3
"""Per-signature census of the convolutions of one 1080p GOP-16 encode: launches, time, TFLOP/s (HIP events on one
stream).  CENSUS_SCHEDULE=pairs: the harness loop pair by pair through stream launches (the shapes the default path
replays from its launch plans); default: the stage-batched schedule."""
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
pairs = os.environ.get("CENSUS_SCHEDULE") == "pairs"
net.engine().use_graphs = False
run = (lambda: pmctf_gop.encode_gop(net, frames, H, W, 3, tmp)) if pairs else (lambda: pmctf_gop.encode_gop_batched(net, frames, H, W, 3, tmp))
import gc, time
gc_log = []                                 # (start, seconds, generation) of every collection during the probed run
def _gc_cb(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        gc_log.append((_t[0], time.perf_counter() - _t[0], info["generation"]))
host_t = []                                 # host clock when each probed convolution was matched (just before its launch)
_match = match
def match(conv, x, stride):
    host_t.append(time.perf_counter())
    return _match(conv, x, stride)
with torch.no_grad():
    run(); torch.cuda.synchronize()
    probe = {"match": match, "events": []}
    ops.CONV_PROBE = probe
    gc.callbacks.append(_gc_cb)
    run(); torch.cuda.synchronize()
    gc.callbacks.remove(_gc_cb)
    ops.CONV_PROBE = None
agg = collections.OrderedDict()
for s, (e0, e1, fl) in zip(sigs, probe["events"]):
    a = agg.setdefault(s, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"total conv ms {tot:.1f}  total TFLOP {sum(a[2] for a in agg.values())/1e12:.1f}")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    (N, h, w, ci), co, k, st = s
    print(f"{N}x{h}x{w} {ci:4d}->{co:4d} k{k} s{st}  n={a[0]:5d} {a[1]:8.1f} ms ({a[1]/tot*100:4.1f}%) {a[1]/a[0]*1e3:8.1f} us  {a[2]/a[1]/1e9:6.1f} TF/s")
if len(sys.argv) > 1:                       # per-launch durations of one signature, in launch order: N H W Cin Cout [K [S]]
    want = tuple(int(v) for v in sys.argv[1:5])
    kk = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    ss = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    hits = [(i, round(e0.elapsed_time(e1) * 1e3)) for i, (s, (e0, e1, fl)) in enumerate(zip(sigs, probe["events"]))
            if s[0] == want and s[1] == int(sys.argv[5]) and s[2] == kk and s[3] == ss]
    print("per-launch us:", [d for _, d in hits])
    med = sorted(d for _, d in hits)[len(hits) // 2]
    print(f"median {med} us; launches above 4x the median, with the convolution probed just before them and the gap between "
          f"that one's end and this one's start (GPU idle or other kernels):")
    for i, d in hits:
        if d > 4 * med and i > 0:
            (pn, ph, pw, pci), pco, pk, pst = sigs[i - 1]
            gap = probe["events"][i - 1][1].elapsed_time(probe["events"][i][0]) * 1e3
            print(f"  launch #{i}: {d} us; before it {pn}x{ph}x{pw} {pci}->{pco} k{pk} s{pst}, gap {gap:.0f} us")
            t0, t1 = host_t[i], host_t[i + 1] if i + 1 < len(host_t) else float("inf")
            print(f"    host: {1e3 * (t1 - t0):.1f} ms between this launch and the next probed one; collections of Python's "
                  f"cyclic GC inside that window: "
                  f"{[(f'gen{g}', f'{1e3 * dt:.1f} ms') for ts, dt, g in gc_log if t0 <= ts <= t1] or 'none'}")
    big = sorted(gc_log, key=lambda r: -r[1])[:5]
    print("longest collections of the probed run:", [(f"gen{g}", f"{1e3 * dt:.1f} ms") for _, dt, g in big])
    import bisect
    print("every collection of Python's cyclic GC that took more than 5 ms, and the probed convolution it interrupted (the "
          "host stops between the start marker and the launch: the GPU idles, the marker pair reads the pause):")
    for ts, dt, g in gc_log:
        if dt > 5e-3:
            i = bisect.bisect_right(host_t, ts) - 1
            if 0 <= i < len(sigs):
                (n_, h_, w_, ci_), co_, k_, st_ = sigs[i]
                e0, e1, _ = probe["events"][i]
                print(f"  gen{g} {1e3 * dt:.1f} ms inside launch #{i}: {n_}x{h_}x{w_} {ci_}->{co_} k{k_} s{st_}, whose markers read "
                      f"{e0.elapsed_time(e1) * 1e3:.0f} us (median of its signature: "
                      f"{sorted(round(a.elapsed_time(b) * 1e3) for s2, (a, b, _) in zip(sigs, probe['events']) if s2 == sigs[i])[len([1 for s2 in sigs if s2 == sigs[i]]) // 2]} us)")
if os.environ.get("CENSUS_SMALL"):
    print("--- planes of at most 70 000 output pixels, by time ---")
    small = [(s, a) for s, a in agg.items() if s[0][0] * (s[0][1] // s[3]) * (s[0][2] // s[3]) <= 70000]
    st = sum(a[1] for _, a in small)
    print(f"small-plane conv ms {st:.1f} of {tot:.1f}")
    for s, a in sorted(small, key=lambda kv: -kv[1][1])[:40]:
        (N, h, w, ci), co, k, stv = s
        print(f"{N}x{h}x{w} {ci:4d}->{co:4d} k{k} s{stv}  n={a[0]:5d} {a[1]:8.1f} ms {a[1]/a[0]*1e3:8.1f} us  {a[2]/a[1]/1e9:6.1f} TF/s")
if os.environ.get("CENSUS_K"):
    kk = int(os.environ["CENSUS_K"])
    rows = [(s, a) for s, a in agg.items() if s[2] == kk]
    print(f"--- all {kk}x{kk} convolutions: {sum(a[1] for _, a in rows):.1f} ms ---")
    for s, a in sorted(rows, key=lambda kv: -kv[1][1]):
        (N, h, w, ci), co, k, stv = s
        print(f"{N}x{h}x{w} {ci:4d}->{co:4d} k{k} s{stv}  n={a[0]:5d} {a[1]:8.1f} ms {a[1]/a[0]*1e3:8.1f} us  {a[2]/a[1]/1e9:6.1f} TF/s")
