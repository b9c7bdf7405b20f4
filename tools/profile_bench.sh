#!/bin/bash
# Profiles `python bench.py --config $2` on the GPU box: kernel trace + stats, then HBM / SQ counters in their own
# passes (gpurun refuses --pmc combined with the API trace domains).  Output: gpurun_out/prof_$1/
#   tools/profile_bench.sh <tag> [config=2] [steps=5]       then: python tools/summarize_profile.py <tag> <name-under-profiles>
set -u
TAG=${1:-r02_cfg2}
CFG=${2:-2}
STEPS=${3:-5}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-serving --no-end-to-end --soak-seconds 0 --corpus-cache /tmp/bftkv_corpus"
python bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o s -- python $R/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
cd $R
# keep only the small summaries (kernel_stats + counter csvs are small; drop big traces)
find $OUT -size +6M -delete
ls $OUT $OUT/trace 2>/dev/null | head -30
