mkdir -p gpurun_out/r3k
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3k/pytest_gpu.txt 2>&1; echo "pytest rc=$?" ; tail -4 gpurun_out/r3k/pytest_gpu.txt | cut -c1-200
python bench.py > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r3k/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3k/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'roofline', j['roofline']['achieved'], j['roofline']['frac'], 'wall', j['bench_wall_s'])
for k in ('stream_launches_single_stream','deferred_store_only','stage_batched','cross_gop_batched'):
    print(k, j.get(k,{}).get('value'))
print('decode', {k:v for k,v in j.get('decode_pair',{}).items() if k in ('h_pair','h_and_l_pair')})
print({k:(v['value'], v['stage_batched']['value'], v['vs_reference_cpu']['frames_with_identical_bits']) for k,v in j.get('aux_profiles',{}).items()})
print(j.get('aux_errors'), j.get('aux_skipped_over_budget'), j['cpu_baseline']['value'])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
