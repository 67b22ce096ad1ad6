mkdir -p gpurun_out
./tools/alubench > gpurun_out/alubench.txt 2>&1; cat gpurun_out/alubench.txt
