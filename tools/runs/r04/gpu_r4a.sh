# round 4, call A: H2D rates of the box, the whole GPU suite (with the new full-size tests), the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/microbench/build/h2d_rates > gpurun_out/h2d_rates.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/pytest_gpu.log 2>&1
tail -30 gpurun_out/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 1500 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
cat gpurun_out/h2d_rates.txt
