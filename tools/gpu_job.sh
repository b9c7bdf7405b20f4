export TMPDIR=/tmp
T=${1:-r03g}
mkdir -p gpurun_out/$T
R=$PWD
timeout 600 python -m pytest tests/test_gpu_protocol.py -m gpu -x -q -k "batcher" > gpurun_out/$T/pytest.log 2>&1; tail -3 gpurun_out/$T/pytest.log
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin 4096 64 > /dev/null 2>&1
for lanes in 3 4; do
  /tmp/batcher_load /tmp/load.bin 256 200 $lanes 8,32,128,256,512 > gpurun_out/$T/lanes$lanes.json 2>&1
done
for f in gpurun_out/$T/lanes*.json; do echo $f; cut -c1-3400 $f; done
bash tools/ab.sh ${T}_ab 2 20 2 mb256.so mb64.so mb128.so 2>&1 | tail -8
