#!/usr/bin/env python3
"""GPU idle time from a rocprofv3 kernel trace: union of the kernels' busy intervals over all streams, total idle, the
largest gaps and the kernels on either side.  usage: idle_gaps.py <dir with *kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
span0, span1 = K[0][0], max(k[1] for k in K)
busy, cur_end, gaps, last = 0, K[0][0], [], K[0]
for s, e, n in K:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - span0, last[2][:60], n[:60]))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = (s, e, n)
print(f"kernels {len(K)}  span {(span1 - span0) / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {(span1 - span0 - busy) / 1e6:.1f} ms "
      f"({(span1 - span0 - busy) / (span1 - span0) * 100:.2f} %)")
small = [g for g in gaps if g[0] < min_gap * 1e3]
print(f"gaps < {min_gap:.0f} us: {len(small)} totalling {sum(g[0] for g in small) / 1e6:.1f} ms; mean {sum(g[0] for g in small) / max(1, len(small)) / 1e3:.2f} us")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0] / 1e3:9.1f} us at {g[1] / 1e6:9.1f} ms   after {g[2]}   before {g[3]}")
