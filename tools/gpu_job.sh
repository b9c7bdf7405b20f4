export TMPDIR=/tmp
timeout 900 python bench.py --config 5 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err; echo "rc=$?"
