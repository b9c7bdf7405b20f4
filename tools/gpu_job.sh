export TMPDIR=/tmp
T=${1:-r03f}
mkdir -p gpurun_out/$T
R=$PWD
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest.log; tail -15 gpurun_out/$T/pytest.log
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin 4096 64 > /dev/null 2>&1
for lanes in 2 3 4; do
  /tmp/batcher_load /tmp/load.bin 256 200 $lanes 1,64,128,256,512 > gpurun_out/$T/lanes$lanes.json 2>&1
done
BFTKV_NO_WIDE8=1 /tmp/batcher_load /tmp/load.bin 256 200 2 1,128 > gpurun_out/$T/lanes2_nowide8.json 2>&1
for f in gpurun_out/$T/lanes*.json; do echo $f; cut -c1-3400 $f; done
