export TMPDIR=/tmp
T=r03m
mkdir -p gpurun_out/$T
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log; tail -8 gpurun_out/$T/pytest.log
rm -rf gpurun_out/prof_r03_cfg2
timeout 600 bash tools/profile_bench.sh r03_cfg2 2 5 > gpurun_out/prof_r03_cfg2.log 2>&1; echo "cfg2 profile rc=$?"
SKIP_TESTS=1 bash tools/gpu_round.sh r03_final
