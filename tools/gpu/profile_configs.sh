# Per-configuration evidence for every kernel family: kernel-trace durations next to the HIP-event ones, HBM traffic from the PMC
# counters, and (round 4) the vector-ALU issue counters -- every counter in a pass of its own, no trace domains mixed in.  Run
# through gpurun; summaries are copied to profiles/ by hand.  BENCH_TWIN=0: the math-free twins of the read rows are launched by the
# plain pass only (the profiling passes cut the dispatch stream by the 'launches' each row prints).
set -x
mkdir -p gpurun_out/prof_cfg
export TMPDIR=/tmp
export BENCH_TWIN=0
export BENCH_SAME=0      # round 5: the rotating (fresh-data) launches only; the one-set pass beside them belongs to the plain pass
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_cfg/kt -o kt --output-format csv -- bash -c "cd $R && python tools/bench_configs.py 2>/dev/null > gpurun_out/prof_cfg/configs_under_trace.jsonl" > $R/gpurun_out/prof_cfg/kt.log 2>&1
for c in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU}; do
  d=$(echo $c | tr A-Z a-z)
  timeout 900 rocprofv3 --pmc $c --kernel-include-regex "write_|read_px" -d $R/gpurun_out/prof_cfg/$d -o p --output-format csv -- bash -c "cd $R && python tools/bench_configs.py > /dev/null 2>&1" > $R/gpurun_out/prof_cfg/$d.log 2>&1
done
cd $R
python tools/summarize_pmc.py gpurun_out/prof_cfg gpurun_out/prof_cfg/configs_under_trace.jsonl gpurun_out/prof_cfg/configs_traffic.json
rm -rf gpurun_out/prof_cfg/kt gpurun_out/prof_cfg/fetch_size gpurun_out/prof_cfg/write_size gpurun_out/prof_cfg/sq_insts_valu gpurun_out/prof_cfg/sq_active_inst_valu   # raw CSVs are large; the summary is what is kept
