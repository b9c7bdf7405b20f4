#!/bin/bash
# kernel trace of one bench config: tools/gpu_trace.sh <config> <out-tag> [extra bench args]
CFG=$1; TAG=$2; shift 2
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/$TAG; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -o t -- python $R/bench.py --config $CFG --no-cpu-baseline --steps 5 --warmup 2 "$@" > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
python - "$R/gpurun_out/$TAG/trace/t_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    print("%-40s calls %4s avg %10.1f us  total %9.2f ms" % (r["Name"].split("(")[0].replace("void ", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
find $R/gpurun_out/$TAG -size +6M -delete
