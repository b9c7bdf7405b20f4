# the job of the moment for `gpurun -- 'bash tools/gpu_job.sh'` (edited per experiment; this is the round's closing form:
# the GPU suite, smoke(), the driver-shaped bench line)
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/final/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/final/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/final/bench_full.json > gpurun_out/final/line.json 2> gpurun_out/final/stderr.txt; echo rc=$? bytes=$(wc -c < gpurun_out/final/line.json)
