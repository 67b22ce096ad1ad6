# round 2, first GPU pass: the refactored library (contexts + workers), new multi-device tests, whole suite, PCIe-path numbers
mkdir -p gpurun_out
timeout 300 python -X faulthandler -m pytest tests/test_gpu_multidevice.py tests/test_gpu_host_shim.py -m gpu -q -x 2>&1 | tail -15
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15
timeout 300 python tools/bench_pcie.py > gpurun_out/bench_pcie.jsonl 2> gpurun_out/bench_pcie.err; cat gpurun_out/bench_pcie.jsonl; tail -3 gpurun_out/bench_pcie.err
timeout 300 python tools/bench_host_shim.py > gpurun_out/host_shim.jsonl 2>gpurun_out/host_shim.err; cut -c1-260 gpurun_out/host_shim.jsonl; tail -3 gpurun_out/host_shim.err
timeout 200 python bench.py --steps 100 --no-cpu-baseline 2>&1 | tail -2
