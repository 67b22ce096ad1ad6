#!/bin/bash
# Round 4: the tables of the round on the final library -- per-row bench (+ read twins) + kernel trace + PMC traffic and VALU issue (r04f),
# then the headline profile (profile_r04: kernel trace cut to the timed region, FETCH_SIZE / WRITE_SIZE passes, the plain bench line, smoke)
bash tools/gpu/configs_r04.sh > gpurun_out/r04f.log 2>&1; tail -5 gpurun_out/r04f.log
bash tools/gpu/profile_r04.sh > gpurun_out/profile_r04.log 2>&1; tail -12 gpurun_out/profile_r04.log | cut -c1-1500
