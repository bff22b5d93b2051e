set -e
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp
PMCTF_MULTI_STREAM=1 rocprofv3 --kernel-trace -d /tmp/tr -o eager --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/eager_gop.py 1 > $GRAFT_REPO_ROOT/gpurun_out/r3a/trace_run_ms.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/idle_gaps.py /tmp/tr 200 -3330 480 3 > gpurun_out/r3a/idle_gaps_eager_ms.txt 2>&1
