export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -6
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
bash tools/profile_bench.sh r03g_cfg2 2 20 2>&1 | tail -1
python tools/summarize_profile.py r03g_cfg2 r03_cfg2 2>&1 | tail -1
