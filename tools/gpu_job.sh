export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_gpu_protocol.py -m gpu -x -q -k "other_sizes or outside_the_keyring" 2>&1 | tail -15
for k in dsa1024 dsa2048 dsa3072 dsa1024,dsa3072,dsa1536,dsa2048; do
  timeout 600 python tools/dsa_rate.py --json --replicas 16 --dsa-fraction 1.0 --dsa-kind $k --bits 14 --writes 600 --tile 24 2>/dev/null | tail -1
done > gpurun_out/r06h/dsa_rates_by_group_size.jsonl
cat gpurun_out/r06h/dsa_rates_by_group_size.jsonl
bash tools/profile_bench.sh r06_cfg4 4 2 > /dev/null 2>&1
bash tools/profile_bench.sh r06_cfg5 5 16 > /dev/null 2>&1
ls gpurun_out/prof_r06_cfg4 gpurun_out/prof_r06_cfg5 | head -30
