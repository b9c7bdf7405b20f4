export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for a in "--steps 20 --warmup 2" "--steps 20 --warmup 5" "--steps 5 --warmup 1"; do
python bench.py --gpus 1 $a --no-cpu-baseline --no-serving --soak-seconds 0 --corpus-cache /tmp/c2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$a  ms/step %.3f value %.1fM  rsa %.3f sf %.3f'%(d['ms_per_step'], d['value']/1e6, d['kernel_ms']['k_rsa_modexp'], d['kernel_ms']['single_flight']['rsa']))"
done
