# rocprofv3 counter passes (one per counter group, never combined with sys/hip traces) for ONE kernel of a
# micro-benchmark.  usage: pmc_kernel.sh <tag> <kernel-name regex> <script.py> [args...]   (run on the GPU box)
# The profiled program is ALWAYS `python3 <script.py> ...`, started by rocprofv3 itself: with --pmc the profiler's preloaded
# library initialises the GPU before the program starts, so any hop after `--` that re-executes (env, bash -c, a
# `#!/usr/bin/env` script, a launcher) is an exec from a process that holds the GPU — which this pool forbids.
# writes gpurun_out/pmc/<tag>/pmc_<group>.csv (+ trace_<group>.csv: the kernel's durations in the same pass)
set -e
tag=$1; shift; pat=$1; shift
case "$1" in *.py) ;; *) echo "usage: $0 <tag> <kernel regex> <script.py> [args...] (the script runs under python3)"; exit 2;; esac
script=$(realpath "$1"); shift; set -- "$script" "$@"     # the passes run from /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  g=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_$g
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$g -- python3 "$@" > $out/$g.log 2>&1
  f=$(find /tmp/pmc_$g -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "$pat" $f) > $out/pmc_$g.csv
  k=$(find /tmp/pmc_$g -name "*kernel_trace.csv" | head -1)
  (head -1 $k; grep -E "$pat" $k) > $out/trace_$g.csv
  rm -rf /tmp/pmc_$g
done
timeout -k 10 240 python3 "$@" > $out/plain.log 2>&1
wc -l $out/pmc_*.csv
