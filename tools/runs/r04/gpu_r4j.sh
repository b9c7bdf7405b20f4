# round 4, call J: the tests behind the failure of call I onward, then the rocprofv3 passes of the cfg-2 command
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_protocol.py tests/test_gpu_threshold.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_cfg2_full_size_identity_against_the_c_oracle ) > gpurun_out/pytest_gpu_j.log 2>&1
tail -12 gpurun_out/pytest_gpu_j.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
bash tools/profile_bench.sh r04_cfg2 2 5 > gpurun_out/profile_r04_cfg2.log 2>&1
tail -5 gpurun_out/profile_r04_cfg2.log
ls gpurun_out/prof_r04_cfg2 gpurun_out/prof_r04_cfg2/trace 2>/dev/null | head -30
