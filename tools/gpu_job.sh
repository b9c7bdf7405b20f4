export TMPDIR=/tmp
run() { python bench.py --config $1 --steps $2 --warmup 3 --inflight $3 --no-cpu-baseline --no-serving --soak-seconds 0 --corpus-cache /tmp/cc 2>gpurun_out/b.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('big=$BFTKV_BIG_STREAM cfg $1 inflight $3 steps $2  ms/step %.3f value %.1fM  rsa %.3f span %.2f'%(d['ms_per_step'], d['value']/1e6, d['kernel_ms']['k_rsa_modexp'], d['kernel_ms'].get('step_device_span', d['kernel_ms'].get('call_device_span',0))))
" || tail -3 gpurun_out/b.err; }
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for r in 1; do
export BFTKV_BIG_STREAM=0; run 2 300 2; run 2 300 3
export BFTKV_BIG_STREAM=1; run 2 300 2; run 2 300 3
done
