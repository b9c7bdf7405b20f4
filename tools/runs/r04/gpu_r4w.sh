# round 4, call W: the GPU suite twice more on the final tree (flake hunt), new concurrency tests five times
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu_w$r.log 2>&1
tail -4 gpurun_out/pytest_gpu_w$r.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
done
for r in 1 2 3 4 5; do
timeout 600 python -m pytest tests/test_gpu_protocol.py tests/test_gpu_host_pipeline.py -m gpu -x -q -k "cert_verify or three_threads or fails_closed or micro_batcher" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -1
done
