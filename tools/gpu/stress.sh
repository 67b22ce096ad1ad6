# repeat the GPU suite N times (default 12) to catch intermittent crashes; prints the python stack of a crash (faulthandler)
#   bash tools/gpu/stress.sh [N]
for i in $(seq 1 ${1:-12}); do
  timeout 600 python -X faulthandler -m pytest tests -m gpu -q -x -p no:cacheprovider $STRESS_ARGS > gpurun_out/stress_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/stress_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -n "Fatal Python error\|File \"\|Current thread\|Thread 0x" gpurun_out/stress_$i.log | head -40; fi
done
