#!/bin/bash
# Round 5: the tables of the round on the final library -- per-row bench (+ read twins) + kernel trace + PMC traffic and VALU issue (r05f),
# then the headline profile (profile_r05: kernel trace cut to the timed region, FETCH_SIZE / WRITE_SIZE passes, the plain bench line, smoke)
bash tools/gpu/configs_r05.sh > gpurun_out/r05f.log 2>&1; tail -5 gpurun_out/r05f.log
bash tools/gpu/profile_r05.sh > gpurun_out/profile_r05.log 2>&1; tail -12 gpurun_out/profile_r05.log | cut -c1-1500
