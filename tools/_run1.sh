mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "decoder_round" > gpurun_out/r3m/pytest.txt 2>&1; echo "pytest rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/trd -o dec --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/time_decode.py > $GRAFT_REPO_ROOT/gpurun_out/r3m/trace_dec.txt 2>&1
cd $GRAFT_REPO_ROOT
grep decoding gpurun_out/r3m/trace_dec.txt
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/tmp/trd/dec_kernel_trace.csv')) if 'll_ar' in r['Kernel_Name']]
for r in rows: print(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, 'ms')
PY
