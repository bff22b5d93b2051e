mkdir -p gpurun_out/r3h
python bench.py --steps 2 --inflight 2 --no_aux --no_cpu_baseline > gpurun_out/r3h/bench_inflight2.json 2> gpurun_out/r3h/bench_inflight2.err; echo "rc=$?"; tail -3 gpurun_out/r3h/bench_inflight2.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3h/bench_inflight2.json') if l.startswith('{')][-1])
print('inflight2 value', j['value'], j['parity_vs_reference_cpu'])
PY
python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "pair_plan or literal_harness or deferred or decoder" > gpurun_out/r3h/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3h/pytest.txt | cut -c1-200
