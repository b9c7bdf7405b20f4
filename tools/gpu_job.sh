export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dsa or cfg3_shape or gpg" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r06b/dsa_tests.txt
cat gpurun_out/r06b/dsa_tests.txt
