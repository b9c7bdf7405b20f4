export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_protocol.py -m gpu -x -q -k "micro_batcher" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
