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
wave = [r for r in rows if "conv3x3s1_wave_kernel<7, 2>" in r["Kernel_Name"]]
cut = [r for r in wave if r["Grid_Size_X"] == "522240"]          # 2040 workgroups: whole rounds, remainder launched apart
whole = [r for r in wave if r["Grid_Size_X"] == "552960"]        # 2160 workgroups: the whole plane (launch plans)
rem = [r for r in rows if "conv3x3s1_pipe_kernel<7, 1>" in r["Kernel_Name"] and r["Grid_Size_X"] == "122880"]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
def line(name, part):
    ds = [d(r) for r in part]
    if ds:
        print(f"  {name}: {len(ds)} launches, mean {statistics.mean(ds):.1f} us, median {statistics.median(ds):.1f} us, min {min(ds):.1f}, max {max(ds):.1f}")
print(f"kernel trace of: python3 bench.py --steps 1 --no_aux --no_cpu_baseline   ({len(rows)} kernel launches)")
print("dominant convolution, 3x3 112->112 on ONE 576x960 plane")
n = 57 * 16                # the roofline pass codes one GOP through stream launches: 57 such convolutions per coded luma frame
line("stream launches, first pair of every configuration in the warm-up GOP (whole rounds: conv3x3s1_wave_kernel<7,2> grid 2040x256)", cut[:-n])
line("stream launches, roofline pass (the launches bench.py brackets with HIP events)", cut[-n:])
line("   + their remainders (conv3x3s1_pipe_kernel<7,1> grid 480x256), roofline pass", rem[-n:])
line("launch plans, timed GOP (one launch over the whole plane, grid 2160x256; chroma's kernels share the GPU)", whole)
PY
cat $out/dominant_kernel_trace.txt; tail -1 $out/bench_under_rocprof.json | cut -c1-300
