export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
run() { echo "== $*"; env "$@" timeout 300 python tools/hostbuf_rate.py 2>/dev/null | tail -1; }
{
run BFTKV_HB_COPIERS=2
run BFTKV_HB_COPIERS=2 BFTKV_HB_PIECES=4
run BFTKV_HB_PIECES=2
run GPU_MAX_HW_QUEUES=8
} > gpurun_out/r06d/hostbuf_ab4.txt 2>&1
cat gpurun_out/r06d/hostbuf_ab4.txt
