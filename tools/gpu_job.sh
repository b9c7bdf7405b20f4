export TMPDIR=/tmp
mkdir -p gpurun_out/r06g
python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r06g/bench_full.json > gpurun_out/r06g/line.json 2> gpurun_out/r06g/stderr.txt; echo rc=$? bytes=$(wc -c < gpurun_out/r06g/line.json)
bash tools/profile_bench.sh r06_cfg2 2 5 > /dev/null 2>&1
bash tools/profile_bench.sh r06_cfg3 3 3 > /dev/null 2>&1
ls gpurun_out/prof_r06_cfg2 gpurun_out/prof_r06_cfg3
