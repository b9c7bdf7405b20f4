export TMPDIR=/tmp
run() { python bench.py --config $1 --steps $2 --warmup 2 --inflight $3 --no-cpu-baseline --soak-seconds 0 --corpus-cache /tmp/cc 2>gpurun_out/b_$1_$3.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $1 inflight $3 steps $2  ms/step %.3f value %.1fM  kernel_ms %s int_mac %.3f'%(d['ms_per_step'], d['value']/1e6, {k:round(v,2) for k,v in d['kernel_ms'].items() if isinstance(v,(int,float))}, d['int_mac']['frac']))
"; }
run 3 12 1; run 3 12 3; run 3 12 2
run 4 3 1; run 4 3 2
