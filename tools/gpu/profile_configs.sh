# Per-configuration evidence for every kernel family: kernel-trace durations next to the HIP-event ones, and HBM traffic from
# the PMC counters (separate passes, no trace domains mixed in).  Run through gpurun; summaries are copied to profiles/ by hand.
set -x
mkdir -p gpurun_out/prof_cfg
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_cfg/kt -o kt --output-format csv -- bash -c "cd $R && python tools/bench_configs.py 2>/dev/null > gpurun_out/prof_cfg/configs_under_trace.jsonl" > $R/gpurun_out/prof_cfg/kt.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "write_|read_px" -d $R/gpurun_out/prof_cfg/fetch_size -o f --output-format csv -- bash -c "cd $R && python tools/bench_configs.py > /dev/null 2>&1" > $R/gpurun_out/prof_cfg/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "write_|read_px" -d $R/gpurun_out/prof_cfg/write_size -o w --output-format csv -- bash -c "cd $R && python tools/bench_configs.py > /dev/null 2>&1" > $R/gpurun_out/prof_cfg/write.log 2>&1
cd $R
python tools/summarize_pmc.py gpurun_out/prof_cfg gpurun_out/prof_cfg/configs_under_trace.jsonl gpurun_out/prof_cfg/configs_traffic.json
rm -rf gpurun_out/prof_cfg/kt gpurun_out/prof_cfg/fetch_size gpurun_out/prof_cfg/write_size   # raw CSVs are large; the summary is what is kept
