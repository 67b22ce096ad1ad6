#!/bin/bash
# (round 6: bench.py sets the probe's shape through avifgpu_probe_set_shape and reports all six shapes itself -- pattern_shapes_ms; the two
#  environment variables below are no longer read by the library.  Kept as the recipe of profiles/r05/probe_shapes_and_kernel_shapes.txt.)
# Round 5: the headline kernel's math-free twin from workgroups of 1 / 2 / 4 waves, as buffer accesses (the kernels' form) and as
# global_load / global_store with 64-bit lane addresses (tools/membench_r02's form), against membench_r02 itself on the same box.
# Question: membench's bare C4 pattern runs at 0.80 of 8 TB/s from 128-thread workgroups and 0.72 from 256 -- is that reachable by
# the library's pattern (and so by its kernel), and which of the differences (workgroup size, addressing form) carries it?
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu/r05_probe_shapes.sh'      -> gpurun_out/r05g/probe_shapes.txt
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O
F=$O/probe_shapes.txt; : > $F
hipcc --offload-arch=gfx950 -O3 tools/membench_r02.hip -o /tmp/membench_r02 2>/dev/null
for pass in 1 2; do
  for g in 0 1; do for w in 4 2 1; do
    AVIFGPU_PROBE_WAVES=$w AVIFGPU_PROBE_GLOBAL=$g timeout 120 python bench.py --no-cpu-baseline --no-cold --no-pcie --no-c5 --no-live-traffic --steps 120 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)['roofline']
        print('pass $pass  waves $w  %-6s twin %.1f GB/s = %.3f of 8 TB/s | kernel %.3f' % ('global' if $g else 'buffer', r['peak_measured'], r['peak_measured'] / 8000, r['frac']))
" >> $F
  done; done
  timeout 120 /tmp/membench_r02 rotate 2>&1 | head -2 | tail -1 >> $F
  timeout 120 /tmp/membench_r02 patterns 2>&1 | head -4 >> $F
done
cat $F
