mkdir -p gpurun_out
for h in 256 1024 2048; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --height $h --sweep 0x7,0x5,0x3 2>&1 >/dev/null | grep sweep | sed "s/^/h=$h /"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --height $h --transfer clip --sweep 0x7,0x5 2>&1 >/dev/null | grep sweep | sed "s/^/h=$h clip /"
done > gpurun_out/small_sweep.txt 2>&1
cat gpurun_out/small_sweep.txt
