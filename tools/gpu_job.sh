export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
bash tools/profile_bench.sh r03h_cfg2 2 40 2>&1 | tail -1
python tools/summarize_profile.py r03h_cfg2 r03_cfg2 2>&1 | tail -1
python tools/step_timeline.py gpurun_out/prof_r03h_cfg2/trace/t_kernel_trace.csv > gpurun_out/prof_r03h_cfg2/timeline.txt 2>&1; tail -5 gpurun_out/prof_r03h_cfg2/timeline.txt
