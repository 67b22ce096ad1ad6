mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex "read_px|write_px" -d $R/gpurun_out/prof2/sq -o s --output-format csv -- bash -c "cd $R && python tools/bench_configs.py" > $R/gpurun_out/prof2/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --kernel-include-regex "read_px|write_px" -d $R/gpurun_out/prof2/lds -o l --output-format csv -- bash -c "cd $R && python tools/bench_configs.py" > $R/gpurun_out/prof2/lds.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "read_px|write_px" -d $R/gpurun_out/prof2/fetch -o f --output-format csv -- bash -c "cd $R && python tools/bench_configs.py" > $R/gpurun_out/prof2/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "read_px|write_px" -d $R/gpurun_out/prof2/write -o w --output-format csv -- bash -c "cd $R && python tools/bench_configs.py" > $R/gpurun_out/prof2/write.log 2>&1
ls -la $R/gpurun_out/prof2/*/
