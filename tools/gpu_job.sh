export TMPDIR=/tmp
run() { BFTKV_DSA_WBITS=$1 python bench.py --config 3 --steps 12 --warmup 2 --inflight $2 --no-cpu-baseline --soak-seconds 0 --corpus-cache /tmp/cc 2>gpurun_out/b3.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg 3 wbits=$1 inflight $2  ms/step %.3f value %.1fM  kernel_ms %s'%(d['ms_per_step'], d['value']/1e6, {k:round(v,2) for k,v in d['kernel_ms'].items() if isinstance(v,(int,float))}))
" || tail -5 gpurun_out/b3.err; }
run 16 3; run 18 3; run 16 1; run 18 1; run 16 3; run 18 3
