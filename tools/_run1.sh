mkdir -p gpurun_out/r3c
python tools/eager_gop.py 3 > gpurun_out/r3c/eager_pools.txt 2>&1; tail -2 gpurun_out/r3c/eager_pools.txt
python -m pytest tests/ -x -q -m gpu > gpurun_out/r3c/pytest_gpu.txt 2>&1; echo "pytest rc=$?" ; tail -15 gpurun_out/r3c/pytest_gpu.txt | cut -c1-300
