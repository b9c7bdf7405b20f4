export TMPDIR=/tmp
for f in 3 4 3 4; do
python bench.py --steps 300 --warmup 4 --inflight $f --no-cpu-baseline --no-serving --soak-seconds 0 --corpus-cache /tmp/c2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('inflight $f  ms/step %.3f value %.1fM  rsa %.3f sf %.3f'%(d['ms_per_step'], d['value']/1e6, d['kernel_ms']['k_rsa_modexp'], d['kernel_ms']['single_flight']['rsa']))"
done
