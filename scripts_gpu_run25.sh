mkdir -p gpurun_out
timeout 300 ./tools/divcheck_unpremul_f | tee gpurun_out/divcheck_unpremul_f.txt
