set -e
mkdir -p gpurun_out/r3b
PMCTF_LUMA_PRIORITY=-1 PMCTF_HACK_NOJOIN=1 python tools/eager_gop.py 3 > gpurun_out/r3b/eager_hack.txt 2>&1; tail -1 gpurun_out/r3b/eager_hack.txt
PMCTF_LUMA_PRIORITY=-1 python tools/eager_gop.py 3 > gpurun_out/r3b/eager_prio2.txt 2>&1; tail -1 gpurun_out/r3b/eager_prio2.txt
