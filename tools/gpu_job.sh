export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
run() { echo "== $*"; env "$@" timeout 300 python tools/hostbuf_rate.py 2>&1 | tail -1; }
{
run A=1
run HOSTBUF_PINNED=1
run HOSTBUF_PINNED=1 BFTKV_HB_PIECES=5
run HOSTBUF_PINNED=1 BFTKV_HB_PIECES=4
} > gpurun_out/r06d/hostbuf_ab6.txt 2>&1
cat gpurun_out/r06d/hostbuf_ab6.txt
