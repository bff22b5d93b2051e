# rocprofv3 counter passes that say where the waves of ONE kernel spend their cycles (issue / wait / LDS / VMEM), one
# group per pass.  usage: pmc_stalls.sh <tag> <kernel-name regex> <script.py> [args...]   (run on the GPU box)
# The profiled program is ALWAYS `python3 <script.py> ...`, started by rocprofv3 itself: with --pmc the profiler's preloaded
# library initialises the GPU before the program starts, so any hop after `--` that re-executes (env, bash -c, a
# `#!/usr/bin/env` script, a launcher) is an exec from a process that holds the GPU — which this pool forbids.
tag=$1; shift; pat=$1; shift
case "$1" in *.py) ;; *) echo "usage: $0 <tag> <kernel regex> <script.py> [args...] (the script runs under python3)"; exit 2;; esac
script=$(realpath "$1"); shift; set -- "$script" "$@"     # the passes run from /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*" | sort -u > $out/avail.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_FLAT" \
           "SQ_IFETCH SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_WAVES SQ_INSTS_VMEM_RD"; do
  g=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_$g
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$g -- python3 "$@" > $out/$g.log 2>&1
  f=$(find /tmp/pmc_$g -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "$pat" $f) > $out/pmc_$g.csv; else echo "no csv for $g"; tail -2 $out/$g.log; fi
  rm -rf /tmp/pmc_$g
done
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(sys.argv[1] + "/pmc_*.csv"):
    for r in csv.DictReader(open(p)):
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c in sorted(acc):
    v = list(acc[c].values())
    print(f"{c:34s} {sum(v[1:]) / max(1, len(v) - 1):16.0f}   ({len(v)} launches)")
PY
