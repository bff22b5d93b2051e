#!/usr/bin/env python3
"""GPU idle time from a rocprofv3 kernel trace: union of the kernels' busy intervals over all streams, total idle, the
largest gaps and the kernels on either side; gap totals grouped by the kernel that FOLLOWS the gap; and a time-binned
busy profile of a window.  usage: idle_gaps.py <dir with *kernel_trace.csv> [min_gap_us] [win_start_ms win_len_ms bin_ms]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
span0, span1 = K[0][0], max(k[1] for k in K)
busy, cur_end, gaps, last = 0, K[0][0], [], K[0]
for s, e, n in K:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end - span0, short(last[2]), short(n)))
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = (s, e, n)
print(f"kernels {len(K)}  span {(span1 - span0) / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms  idle {(span1 - span0 - busy) / 1e6:.1f} ms "
      f"({(span1 - span0 - busy) / (span1 - span0) * 100:.2f} %)")
small = [g for g in gaps if g[0] < min_gap * 1e3]
print(f"gaps < {min_gap:.0f} us: {len(small)} totalling {sum(g[0] for g in small) / 1e6:.1f} ms; mean {sum(g[0] for g in small) / max(1, len(small)) / 1e3:.2f} us")
hist = collections.Counter()
for g in small:
    hist[min(7, int(g[0] / 1e3).bit_length())] += g[0]
print("small-gap time by size (us):", {f"<{1 << b}": round(v / 1e6, 1) for b, v in sorted(hist.items())}, "ms")
for title, idx in (("followed by", 3), ("preceded by", 2)):
    by = collections.defaultdict(lambda: [0, 0])
    for g in small:
        by[g[idx]][0] += g[0]; by[g[idx]][1] += 1
    print(f"small gaps {title}:")
    for n, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {t / 1e6:8.2f} ms  {c:6d} gaps  {n}")
for g in sorted(gaps, reverse=True)[:16]:
    print(f"  {g[0] / 1e3:9.1f} us at {g[1] / 1e6:9.1f} ms   after {g[2]}   before {g[3]}")
if len(sys.argv) > 5:
    w0, wl, bn = (float(x) * 1e6 for x in sys.argv[3:6])
    if w0 < 0:
        w0 += span1 - span0
    nb = int(wl / bn)
    busy_b, names = [0.0] * nb, [collections.Counter() for _ in range(nb)]
    cnt = [0] * nb
    for s, e, n in K:
        s, e = s - span0 - w0, e - span0 - w0
        if e <= 0 or s >= wl:
            continue
        b0 = max(0, int(s // bn))
        cnt[min(nb - 1, b0)] += 1
        for b in range(b0, min(nb - 1, int(e // bn)) + 1):
            ov = min(e, (b + 1) * bn) - max(s, b * bn)
            if ov > 0:
                busy_b[b] += ov; names[b][short(n)] += ov
    print(f"window from {w0 / 1e6:.1f} ms, bins of {bn / 1e6:.2f} ms: busy % (sum over streams), launches, top kernel")
    for b in range(nb):
        top = names[b].most_common(1)
        print(f"  {b * bn / 1e6:7.1f}  {busy_b[b] / bn * 100:6.1f} %  {cnt[b]:5d}  {top[0][0] if top else '-'}")
