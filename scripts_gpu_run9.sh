mkdir -p gpurun_out
python tools/bench_configs.py > gpurun_out/configs.jsonl 2> gpurun_out/configs.err; cat gpurun_out/configs.jsonl; tail -3 gpurun_out/configs.err
