# Recipe used for profiles/r01: run on the GPU box through gpurun ("bash tools/gpu/profile_c4.sh"), then copy the summaries from gpurun_out/prof into profiles/.
set -x
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt --output-format csv -- bash -c "cd $R && $P > gpurun_out/prof/bench_under_kernel_trace.json" > $R/gpurun_out/prof/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "write_" -d $R/gpurun_out/prof/fetch -o f --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "write_" -d $R/gpurun_out/prof/write -o w --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "write_" -d $R/gpurun_out/prof/sq -o s --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex "write_" -d $R/gpurun_out/prof/lds -o l --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/lds.log 2>&1
cd $R
python bench.py --steps 200 --warmup 20 --pcie > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; cat gpurun_out/bench_r01.json
python tools/bench_configs.py 2>/dev/null > gpurun_out/configs.jsonl
python -c "import __graft_entry__ as g; g.smoke()"
