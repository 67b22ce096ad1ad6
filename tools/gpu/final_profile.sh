mkdir -p gpurun_out
bash tools/gpu/profile_r02.sh > gpurun_out/profile_r02.log 2>&1; grep -E "timed-region average|^\{'workload'|smoke ok" gpurun_out/profile_r02.log; cut -c1-400 gpurun_out/prof/bench_r02.json
bash tools/gpu/profile_configs.sh > gpurun_out/profile_configs.log 2>&1; tail -2 gpurun_out/profile_configs.log
python tools/bench_pcie.py > gpurun_out/bench_pcie_final.jsonl 2>/dev/null; cut -c100-420 gpurun_out/bench_pcie_final.jsonl
python tools/bench_host_shim.py > gpurun_out/host_shim_final.jsonl 2>/dev/null; cut -c1-330 gpurun_out/host_shim_final.jsonl
