export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "dsa or cfg3 or golden or read_answers or mixed" > gpurun_out/pytest_dsa.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_dsa.log | tail -5
run() { BFTKV_DEBUG_DSA=1 python bench.py --config 3 --steps 12 --warmup 2 --inflight 3 --no-cpu-baseline --soak-seconds 0 --corpus-cache /tmp/cc 2>gpurun_out/b3.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg 3 wbits=$1  ms/step %.3f value %.1fM  kernel_ms %s'%(d['ms_per_step'], d['value']/1e6, {k:round(v,2) for k,v in d['kernel_ms'].items() if isinstance(v,(int,float))}))
" || tail -5 gpurun_out/b3.err; grep "dsa tables" gpurun_out/b3.err | tail -1; }
run default
