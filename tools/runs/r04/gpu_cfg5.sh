#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_threshold.py -m gpu -q 2>&1 | tail -2
for P in 0 2 4; do
  BFTKV_MULTIEXP_PARTS=$P python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('parts=$P ms/step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms'].items()})"
done
BFTKV_MULTIEXP_PARTS=4 timeout 600 python -m pytest tests/test_gpu_threshold.py -m gpu -q 2>&1 | tail -2
