#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_threshold.py tests/test_gpu_protocol.py -m gpu -q 2>&1 | tail -3
CONFIGS="2 5" SKIP_TESTS=1 tools/gpu_round.sh g9
python bench.py --inflight 2 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/g9/inflight2.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/g9/inflight2.json').read().strip().splitlines()[-1]); print('inflight2 ms/step', d['ms_per_step'], d['value'])"
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g9/trace -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
cut -c1-90 $R/gpurun_out/g9/trace/t_kernel_stats.csv | head -22
