export TMPDIR=/tmp
mkdir -p gpurun_out/r06j
for x in library torch; do
BFTKV_BENCH_EXCHANGE=$x timeout 600 python bench.py --config 2 --no-serving --no-end-to-end --no-cpu-baseline --soak-seconds 0 --steps 50 --full-json gpurun_out/r06j/cfg2_$x.json > gpurun_out/r06j/cfg2_${x}_line.json 2>/dev/null; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r06j/cfg2_${x}_line.json')); print(d['ms_per_step'], d['value'], d['exchange'], d['verdicts_match_construction'])"
done
BFTKV_FORCE_RCCL=1 BFTKV_BENCH_EXCHANGE=torch timeout 600 python bench.py --config 4 --no-cpu-baseline --soak-seconds 0 --steps 1 --warmup 1 --full-json gpurun_out/r06j/cfg4_torch.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg4', d['ms_per_step'], d['value'], d.get('exchange'), d.get('verdicts_match_construction'))"
