import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id"), r.get("Grid_Size_X", r.get("Grid_Size"))) for r in rows]
K.sort()
big = [k for k in K if "wave_kernel<4" in k[2] and k[1] - k[0] > 9e6]
print(len(big), "big 64->64 launches")
for s, e, nm, q, g in big:
    ov = [(k[2][:60], k[3], (min(e, k[1]) - max(s, k[0])) / 1e3) for k in K if k[0] < e and k[1] > s and (k[0], k[1]) != (s, e)]
    tot = sum(o[2] for o in ov)
    names = {}
    for n_, q_, d in ov: names[(n_, q_)] = names.get((n_, q_), 0) + d
    top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
    print(f"dur {(e-s)/1e3:8.0f} us q{q} grid {g}  overlapped-by {len(ov)} kernels, {tot:8.0f} us:", [(k[0][:40], k[1], round(v)) for k, v in top])
