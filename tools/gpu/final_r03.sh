#!/bin/bash
# Round 3, pass t: the tables of the round on the final library -- per-row bench + kernel trace + PMC traffic (r03f), headline profile (profile_r03)
bash tools/gpu/configs_r03.sh > gpurun_out/r03f.log 2>&1; tail -5 gpurun_out/r03f.log
bash tools/gpu/profile_r03.sh > gpurun_out/profile_r03.log 2>&1; tail -12 gpurun_out/profile_r03.log | cut -c1-1500
