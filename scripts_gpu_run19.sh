mkdir -p gpurun_out
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sweep 0x0,0x7,0x0,0x7,0x5,0x17 2> gpurun_out/b19.txt | cut -c1-120; grep sweep gpurun_out/b19.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --width 7680 --height 4320 --sweep 0x0,0x7 2> gpurun_out/b19b.txt | cut -c1-120; grep sweep gpurun_out/b19b.txt
