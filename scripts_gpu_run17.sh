mkdir -p gpurun_out
./tools/divcheck > gpurun_out/divcheck.txt 2>&1; cat gpurun_out/divcheck.txt
