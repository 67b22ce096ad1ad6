mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12
timeout 300 python -m pytest tests/test_gpu_t2_truth.py -m gpu -q -s 2>&1 | grep -E "EOTF|PQ OETF|mismatch|passed|failed" | head -40
bash tools/gpu/profile_r02.sh > gpurun_out/profile_r02.log 2>&1; tail -25 gpurun_out/profile_r02.log
