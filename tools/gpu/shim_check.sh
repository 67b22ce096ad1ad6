#!/bin/bash
# FormatRecord shim after a scheduler / tile-size change: its tests, then floor / write / read timings
out=gpurun_out/shim_check; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_host_shim.py tests/test_gpu_multidevice.py tests/test_cli.py -m gpu -q -x 2>&1 | tail -2 | tee $out/pytest.txt
for m in floor write read; do timeout 300 python tools/bench_host_shim.py $m 2>/dev/null | cut -c1-330; done | tee $out/host_shim.txt
