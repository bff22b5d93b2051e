#!/usr/bin/env python3
"""The motion CODEC of one 1080p pair (engine.motion_code: MV encoder, hyper pair, four-part prior, MV decoder) through
stream launches, for a rocprofv3 kernel trace: which launches make up its latency-bound chain.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/mvt -o mv -- python3 tools/mv_trace.py
then:            python3 tools/mv_trace.py --report /tmp/mvt/mv_kernel_trace.csv"""
import collections, csv, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # the run ends with four identical motion_code calls: find the period from the end
    names = [r["Kernel_Name"] for r in rows]
    n = next(k for k in range(50, len(rows) // 4) if names[-k:] == names[-2 * k:-k] == names[-3 * k:-2 * k])
    part = rows[-n:]
    t0, t1 = int(part[0]["Start_Timestamp"]), int(part[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part)
    print(f"{len(part)} launches, span {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
    agg = collections.OrderedDict()
    for r in part:
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], r["Grid_Size_X"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{a[1] / 1e3:9.1f} us  {a[0]:4d} x  grid {k[1]:>8s}  {k[0]}")
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learned-pmctf_amd"))
import torch
import pmctf_synth
from pMCTF.models.video.pMCTF_L import pMCTF
net = pMCTF(num_me_stages=4).eval()
net.load_state_dict(pmctf_synth.synth_state_dict(net.state_dict(), seed=0), strict=True)
net = net.cuda(); net.update(force=True)
eng = net.engine(); eng.use_graphs = False
frames = [list(pmctf_synth.frames_to_tensors(f, device="cuda")) for f in pmctf_synth.synth_yuv420(1920, 1080, 2)]
pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, 1152 - t.shape[2]), mode="replicate") if t.shape[2] < 1152 else t
ry, cy = pad(frames[0][0]), pad(frames[1][0])
with torch.no_grad():
    mv = eng.motion_estimate(ry, cy)
    dpb = {"mv_feature": None, "ref_mv_y": None}
    r = eng.motion_code(mv, dpb, 0, 3)                 # first pair of a GOP: no context
    ctx = {"mv_feature": r["mv_feature"].permute(0, 3, 1, 2), "ref_mv_y": r["mv_y_hat"].permute(0, 3, 1, 2)}
    torch.cuda.synchronize()
    for _ in range(4):                      # four identical chained calls: the report takes the last quarter of the launches
        eng.motion_code(mv, ctx, 0, 3)
        torch.cuda.synchronize()
print("done")
