#!/bin/bash
# A/B on ONE box: tools/ab.sh <tag> <config> <steps> <rounds> a.so b.so ...   (library variants under variants/, see DESIGN.md "measuring")
# Runs the bench with each variant copied over bftkv_amd/libbftkv_gpu.so, interleaved, and prints ms/step + kernel_ms per run.
TAG=$1; CFG=$2; STEPS=$3; ROUNDS=$4; shift 4
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp bftkv_amd/libbftkv_gpu.so $OUT/_orig.so
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    cp variants/$v bftkv_amd/libbftkv_gpu.so
    python bench.py --config $CFG --steps $STEPS --warmup 2 --inflight 1 --no-cpu-baseline --no-serving --corpus-cache /tmp/abcorpus > $OUT/r${r}_$v.json 2> $OUT/r${r}_$v.err
    python - "$OUT/r${r}_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    km = {k: round(v, 3) for k, v in d.get("kernel_ms", {}).items() if isinstance(v, (int, float))}
    print("%-12s ms/step %.3f  %s  sclk %s" % (sys.argv[2], d["ms_per_step"], km, d.get("int_mac", {}).get("sclk_mhz_in_kernel")))
except Exception as e:
    print(sys.argv[2], "no JSON line:", e)
PY
  done
done
cp $OUT/_orig.so bftkv_amd/libbftkv_gpu.so; rm $OUT/_orig.so
