#!/bin/sh
# cfg-3-shaped runs at larger DSA populations: $1 replicas (half DSA), $2 variables, then the widths to pin ("auto" = policy)
# usage: sh tools/dsa_population.sh 128 5000 auto 8 14 16
n=$1; items=$2; shift 2
mkdir -p gpurun_out /tmp/cc
for w in "$@"; do
  if [ "$w" = auto ]; then unset BFTKV_DSA_WBITS; else export BFTKV_DSA_WBITS=$w; fi
  extra="--no-cpu-baseline"
  [ "$w" = auto ] && extra="--cpu-budget 20"
  python bench.py --config 3 --replicas $n --items $items --steps 6 --warmup 2 --soak-seconds 0 --corpus-cache /tmp/cc/c3 $extra \
      > gpurun_out/r5_cfg3_n${n}_w${w}.json 2> gpurun_out/r5_cfg3_n${n}_w${w}.err || tail -5 gpurun_out/r5_cfg3_n${n}_w${w}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5_cfg3_n${n}_w${w}.json"))
    cb=d.get("cpu_baseline") or {}
    print("n=${n} w=${w}:", d["dsa_tables"], "ms/step %.2f" % d["ms_per_step"], "value %.3g" % d["value"], "int_mac %.3f" % d["int_mac"]["frac"],
          "kernel_ms", {k: round(v,2) for k,v in d["kernel_ms"].items() if isinstance(v,(int,float))}, "identity", {k:v for k,v in cb.items() if "identical" in k}, "match", d.get("verdicts_match_construction"))
except Exception as e:
    print("n=${n} w=${w}: failed", e)
PY
done
