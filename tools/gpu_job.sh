export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_threshold.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
run() { BFTKV_MULTIEXP_LANES=$1 python bench.py --config 5 --steps 20 --warmup 3 --inflight $2 --no-cpu-baseline --soak-seconds 0 2>gpurun_out/b5.err | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('cfg 5 lanes $1 inflight $2  ms/step %.3f value %.2fM ops/s  CalculateR %.2f ms int_mac %.3f'%(d['ms_per_step'], d['value']/1e6, d['kernel_ms']['dsa_calculate_r_2t8_2048_256'], d['int_mac']['frac']))
" || tail -5 gpurun_out/b5.err; }
run 4 1; run 16 1; run 8 1; run 16 2; run 16 4; run 4 4
