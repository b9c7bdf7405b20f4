export TMPDIR=/tmp
T=${1:-r03c}
mkdir -p gpurun_out/$T
grep -m1 "model name" /proc/cpuinfo > gpurun_out/$T/cpu.txt; grep -m1 -o sha_ni /proc/cpuinfo >> gpurun_out/$T/cpu.txt; nproc >> gpurun_out/$T/cpu.txt; cat gpurun_out/$T/cpu.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log; tail -12 gpurun_out/$T/pytest.log
timeout 300 python tools/serving/small_call.py > gpurun_out/$T/small_call.txt 2>&1; tail -2 gpurun_out/$T/small_call.txt
LANES="${LANES:-4 1 2}" timeout 900 bash tools/serving/run.sh > gpurun_out/$T/serving.log 2>&1; tail -4 gpurun_out/$T/serving.log | cut -c1-2500
mkdir -p gpurun_out/$T/serving; cp gpurun_out/serving/*.json gpurun_out/$T/serving/
