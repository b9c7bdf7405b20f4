export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "big_batch_with_other_hashes or golden" ) > gpurun_out/pytest_x.log 2>&1
tail -12 gpurun_out/pytest_x.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
