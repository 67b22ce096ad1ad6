set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"
AVIFGPU_HOT_VARIANT=1 timeout 600 python bench.py --steps 100 --warmup 10 --pcie > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err; cat gpurun_out/bench_v1.json
AVIFGPU_HOT_VARIANT=0 timeout 600 $B > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err; cat gpurun_out/bench_v0.json
AVIFGPU_HOT_VARIANT=1 timeout 600 $B --transfer clip > gpurun_out/bench_v1_clip.json 2>&1; cat gpurun_out/bench_v1_clip.json
AVIFGPU_HOT_VARIANT=0 timeout 600 $B --transfer clip > gpurun_out/bench_v0_clip.json 2>&1; cat gpurun_out/bench_v0_clip.json
timeout 600 $B --chroma 420 > gpurun_out/bench_420.json 2>&1; cat gpurun_out/bench_420.json
P="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt --output-format csv -- bash -c "cd $GRAFT_REPO_ROOT && $P" > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "write_" -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -o f --output-format csv -- bash -c "cd $GRAFT_REPO_ROOT && $P" > $GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "write_" -d $GRAFT_REPO_ROOT/gpurun_out/prof_write -o w --output-format csv -- bash -c "cd $GRAFT_REPO_ROOT && $P" > $GRAFT_REPO_ROOT/gpurun_out/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "write_" -d $GRAFT_REPO_ROOT/gpurun_out/prof_sq -o s --output-format csv -- bash -c "cd $GRAFT_REPO_ROOT && $P" > $GRAFT_REPO_ROOT/gpurun_out/prof_sq.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name '*.csv' | head -30; du -sh gpurun_out
