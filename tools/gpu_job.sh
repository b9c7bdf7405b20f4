export TMPDIR=/tmp
T=${1:-r03k}
mkdir -p gpurun_out/$T
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log; tail -25 gpurun_out/$T/pytest.log
