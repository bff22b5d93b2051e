mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "decoder or still_image or cross_decode or full_size or corrupt" > gpurun_out/r3d/pytest_dec.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3d/pytest_dec.txt
PROFILE=0 python tools/decode_profile.py > gpurun_out/r3d/decode_profile_H.txt 2>&1; grep -v amdgpu gpurun_out/r3d/decode_profile_H.txt
PROFILE=0 python tools/decode_profile.py L > gpurun_out/r3d/decode_profile_L.txt 2>&1; grep -v amdgpu gpurun_out/r3d/decode_profile_L.txt
