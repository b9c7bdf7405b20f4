#!/bin/bash
# tests, every bench config, then the rocprofv3 passes of every config (profiles/<tag>_cfgN_*)
TAG=${1:-r02b}
tools/gpu_round.sh ${TAG}_run
for c in 2 3 4 5; do
  steps=5; [ $c = 4 ] && steps=2; [ $c = 3 ] && steps=3
  tools/profile_bench.sh ${TAG}_cfg$c $c $steps > /dev/null 2>&1
done
ls gpurun_out | grep prof_${TAG}
