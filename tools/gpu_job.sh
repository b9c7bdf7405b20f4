export TMPDIR=/tmp
mkdir -p gpurun_out/r06f
BFTKV_FORCE_RCCL=1 BFTKV_BENCH_EXTRAS_IN_PROCESS=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r06f/rehearsal_full.json > gpurun_out/r06f/rehearsal_line.json 2> gpurun_out/r06f/rehearsal_stderr.txt; echo rehearsal rc=$? bytes=$(wc -c < gpurun_out/r06f/rehearsal_line.json)
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r06f/gpu_suite.txt; cat gpurun_out/r06f/gpu_suite.txt
