export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r06e
python tools/threshold_rate.py --ops 4000 --nodes 64 --k 22 > gpurun_out/r06e/rates_64_after.json 2>/dev/null
python tools/threshold_rate.py --ops 4000 --nodes 256 --k 86 > gpurun_out/r06e/rates_256_after.json 2>/dev/null
python tools/threshold_rate.py --ops 10000 > gpurun_out/r06e/rates_10_after.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06e/trace64b -o t -- python $R/tools/threshold_rate.py --ops 4000 --nodes 64 --k 22 > /dev/null 2>&1
cd $R
find gpurun_out/r06e -size +4M -delete
cat gpurun_out/r06e/rates_64_after.json; cat gpurun_out/r06e/rates_256_after.json
head -8 gpurun_out/r06e/trace64b/t_kernel_stats.csv
timeout 1200 python -m pytest tests/test_gpu_threshold.py -m gpu -x -q 2>&1 | tail -3
