# rocprofv3 kernel trace + stats of the bench command (run on the GPU box): writes the per-kernel stats and, for the
# dominant convolution's launch shape (one 576x960 plane: 2040 workgroups of 256 threads), the durations by phase of the
# run.  usage: bash tools/profile_bench.sh <out dir under gpurun_out>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python3 $R/bench.py --steps 1 --no_aux --no_cpu_baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cp /tmp/prof_bench/bench_kernel_stats.csv $out/bench_kernel_stats.csv
python3 - <<PY > $out/dominant_kernel_trace.txt
import csv, statistics
rows = [r for r in csv.DictReader(open("/tmp/prof_bench/bench_kernel_trace.csv"))]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
dom = [r for r in rows if "conv3x3s1_wave_kernel<7, 2>" in r["Kernel_Name"] and r["Grid_Size_X"] == "522240"]
rem = [r for r in rows if "conv3x3s1_pipe_kernel<7, 1>" in r["Kernel_Name"] and r["Grid_Size_X"] == "122880"]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"kernel trace of: python3 bench.py --steps 1 --no_aux --no_cpu_baseline   ({len(rows)} kernel launches)")
print("dominant convolution, 3x3 112->112 on ONE 576x960 plane: conv3x3s1_wave_kernel<7,2> grid 2040x256 + conv3x3s1_pipe_kernel<7,1> grid 480x256")
n = len(dom) // 3
for name, part in (("warm-up GOP (stream launches while the plans are recorded)", dom[:n]), ("timed GOP (plans replayed, luma and chroma kernels share the GPU)", dom[n:2 * n]), ("roofline pass (stream launches, one stream)", dom[2 * n:])):
    ds = [d(r) for r in part]
    print(f"  {name}: {len(ds)} launches, mean {statistics.mean(ds):.1f} us, median {statistics.median(ds):.1f} us, min {min(ds):.1f}, max {max(ds):.1f}")
ds = [d(r) for r in rem[2 * (len(rem) // 3):]]
print(f"  remainder kernel in the roofline pass: {len(ds)} launches, mean {statistics.mean(ds):.1f} us")
PY
cat $out/dominant_kernel_trace.txt; tail -1 $out/bench_under_rocprof.json | cut -c1-300
