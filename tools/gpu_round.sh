#!/bin/bash
# One GPU-box visit: the -m gpu tests, then every bench config once (short), everything logged under gpurun_out/$1/.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
for cfg in ${CONFIGS-2 3 4 5}; do
  timeout 900 python bench.py --config $cfg --steps ${STEPS:-5} --warmup 2 > $OUT/bench_cfg$cfg.json 2> $OUT/bench_cfg$cfg.err; echo "cfg$cfg rc=$?"
  python - "$OUT/bench_cfg$cfg.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    km = {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items() if isinstance(v, (int, float))}
    cb = d.get("cpu_baseline", {})
    print("  value %.4g %s  ms/step %.3f  kernel_ms %s  identical %s" % (d["value"], d["unit"], d["ms_per_step"], km,
          [v for k, v in cb.items() if "identical" in k]))
except Exception as e:
    print("  no JSON line:", e)
PY
  tail -2 $OUT/bench_cfg$cfg.err | cut -c1-300
done
