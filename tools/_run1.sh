mkdir -p gpurun_out/r3j
for lv in -1 1 0 2; do PMCTF_STAGGER_LEVEL=$lv python tools/eager_gop.py 3 > gpurun_out/r3j/stagger_$lv.txt 2>&1; echo "stagger level $lv: $(tail -1 gpurun_out/r3j/stagger_$lv.txt | cut -c1-60)"; done
