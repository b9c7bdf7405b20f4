export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r06l gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_host_pipeline.py -m gpu -x -q 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout 300 python tools/hostbuf_rate.py 2>/dev/null | tail -1; }
{
run A=1
run BFTKV_HB_PIECES=4
run A=2
} > gpurun_out/r06d/hostbuf_ab7.txt 2>&1
cat gpurun_out/r06d/hostbuf_ab7.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06l/trace2 -o t -- python $R/tools/hostbuf_trace.py > $R/gpurun_out/r06l/timeline2.json 2>/dev/null
cd $R
grep -h "k_expand_segments" gpurun_out/r06l/trace2/*.csv gpurun_out/r06l/trace2/*/*.csv 2>/dev/null | awk -F, '{print ($(NF-1)-$(NF-2))/1000}' | tail -6
find gpurun_out/r06l -size +3M -delete
