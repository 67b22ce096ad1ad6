mkdir -p gpurun_out
./tools/membench > gpurun_out/membench2.txt 2>&1; cat gpurun_out/membench2.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sweep 0x7,0x17,0x5,0x200007,0x800007,0x2000007 2> gpurun_out/b12.txt | cut -c1-330; grep sweep gpurun_out/b12.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --transfer clip --sweep 0x7,0x17,0x5,0x200007,0x800007,0x2000007 2> gpurun_out/b12c.txt | cut -c1-100; grep sweep gpurun_out/b12c.txt
