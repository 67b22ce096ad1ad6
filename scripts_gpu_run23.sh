mkdir -p gpurun_out
./tools/divcheck_premul | tee gpurun_out/divcheck_premul.txt
