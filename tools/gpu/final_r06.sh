#!/bin/bash
# Round 6: the tables of the round on the final library -- per-row bench (+ read twins) + memory-free twins (the AG_MATH_ONLY library: build it
# first with tools/ab_variants.sh write_kernels,read_kernels "-DAG_MATH_ONLY=1" mathonly) + kernel trace + PMC traffic (r06f), then the headline
# profile (profile_r06: kernel trace cut to the timed region, FETCH_SIZE / WRITE_SIZE passes, the plain bench line, smoke)
bash tools/gpu/configs_r06.sh > gpurun_out/r06f.log 2>&1; tail -5 gpurun_out/r06f.log
bash tools/gpu/profile_r06.sh > gpurun_out/profile_r06.log 2>&1; tail -12 gpurun_out/profile_r06.log | cut -c1-1500
