for i in 1 2; do
python tools/bench_configs.py "D12 8192^2 RGB f32 -> 12-bit PQ 4:4:4" 2>/dev/null | cut -c1-160
python tools/bench_configs.py "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" 2>/dev/null | cut -c1-160
python tools/bench_configs.py "D12 8192^2 RGB f32 -> 12-bit PQ 4:4:4" "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4" 2>/dev/null | cut -c1-160
done
