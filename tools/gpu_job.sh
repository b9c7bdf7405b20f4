export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
run() { echo "== $*"; env "$@" timeout 300 python tools/hostbuf_rate.py 2>/dev/null | tail -1; }
{
run A=1
run A=2
run GPU_MAX_HW_QUEUES=8
} > gpurun_out/r06d/hostbuf_ab5.txt 2>&1
cat gpurun_out/r06d/hostbuf_ab5.txt
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -x -q 2>&1 | tail -3
