# round 4, call V: rocprofv3 passes for cfg 4 (2 steps)
export TMPDIR=/tmp
mkdir -p gpurun_out
( time bash tools/profile_bench.sh r04_cfg4 4 2 ) > gpurun_out/profile_r04_cfg4.log 2>&1; tail -3 gpurun_out/profile_r04_cfg4.log
