mkdir -p gpurun_out
AVIFGPU_BENCH_SERIES=1 python bench.py --steps 400 --warmup 0 --no-cpu-baseline 2> gpurun_out/series.txt | cut -c1-200
grep series_ms gpurun_out/series.txt | tr ' ' '\n' | tail -n +2 | awk '{a[NR]=$1} END{for(i=1;i<=NR;i+=20){s=0;for(j=i;j<i+20&&j<=NR;j++)s+=a[j]; printf "%d-%d: %.4f\n", i, i+19, s/20}}'
rocm-smi --showclocks 2>/dev/null | head -20
AVIFGPU_BENCH_SERIES=1 python bench.py --steps 400 --warmup 0 --no-cpu-baseline --transfer clip 2> gpurun_out/series2.txt | cut -c1-100
grep series_ms gpurun_out/series2.txt | tr ' ' '\n' | tail -n +2 | awk '{a[NR]=$1} END{for(i=1;i<=NR;i+=40){s=0;for(j=i;j<i+40&&j<=NR;j++)s+=a[j]; printf "%d-%d: %.4f\n", i, i+39, s/40}}'
