#!/bin/bash
# on the GPU box: build the load generator, make a corpus, sweep caller threads
set -e
R=$PWD
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin ${1:-4096} 64
mkdir -p gpurun_out/serving
/tmp/batcher_load /tmp/load.bin 256 200 | tee gpurun_out/serving/batcher_256_200.json
/tmp/batcher_load /tmp/load.bin 1024 500 | tee gpurun_out/serving/batcher_1024_500.json
