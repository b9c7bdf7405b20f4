export TMPDIR=/tmp
mkdir -p gpurun_out/r06i
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r06i/gpu_suite.txt; cat gpurun_out/r06i/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r06i/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-json gpurun_out/r06i/bench_full.json > gpurun_out/r06i/line.json 2> gpurun_out/r06i/stderr.txt; echo rc=$? bytes=$(wc -c < gpurun_out/r06i/line.json)
for w in 14 16; do BFTKV_DSA_WBITS=$w timeout 600 python bench.py --config 3 --steps 10 --warmup 2 --soak-seconds 0 --no-cpu-baseline --full-json gpurun_out/r06i/cfg3_w$w.json > gpurun_out/r06i/cfg3_w${w}_line.json 2>/dev/null; done
python - <<'PY'
import json
for w in (14,16):
    d=json.load(open('gpurun_out/r06i/cfg3_w%d.json'%w)); print(w, d['ms_per_step'], d['value'], d['dsa_tables'], d['roofline']['launch_ms'], d['int_mac']['frac'])
PY
