export TMPDIR=/tmp
mkdir -p gpurun_out/r06h
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_protocol.py -m gpu -x -q -k "dsa or cfg3 or gpg or other_sizes or cert" 2>&1 | tail -5
for k in dsa2048 dsa3072 dsa1024,dsa3072,dsa1536,dsa2048; do
  timeout 600 python tools/dsa_rate.py --json --replicas 16 --dsa-fraction 1.0 --dsa-kind $k --bits 14 --writes 600 --tile 24 2>/dev/null | tail -1
done > gpurun_out/r06h/dsa_rates_by_group_size_split.jsonl
cat gpurun_out/r06h/dsa_rates_by_group_size_split.jsonl
