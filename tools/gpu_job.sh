export TMPDIR=/tmp
T=${1:-r03e}
mkdir -p gpurun_out/$T
R=$PWD
timeout 900 python -m pytest tests/test_gpu_protocol.py tests/test_c_harness.py -m gpu -x -q > gpurun_out/$T/pytest_protocol.log 2>&1; echo "rc=$?" >> gpurun_out/$T/pytest_protocol.log; tail -4 gpurun_out/$T/pytest_protocol.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batcher or rccl or exchange or with_certificate or transport or message" > gpurun_out/$T/pytest_parity.log 2>&1; tail -4 gpurun_out/$T/pytest_parity.log
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin 4096 64 > /dev/null 2>&1
for lanes in 4 2 8; do
  /tmp/batcher_load /tmp/load.bin 256 200 $lanes 1,64,128,256,512 > gpurun_out/$T/lanes$lanes.json 2>&1
done
GPU_MAX_HW_QUEUES=16 /tmp/batcher_load /tmp/load.bin 256 200 8 64,256 > gpurun_out/$T/lanes8_hwq16.json 2>&1
for f in gpurun_out/$T/lanes*.json; do echo $f; cut -c1-3400 $f; done
