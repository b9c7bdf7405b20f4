#!/bin/bash
# on the GPU box: build the load generator, make a corpus, sweep caller threads and lanes
set -e
R=$PWD
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin ${1:-4096} 64
mkdir -p gpurun_out/serving
for lanes in ${LANES:-4 1 2 8}; do
  /tmp/batcher_load /tmp/load.bin 256 200 $lanes ${THREADS:-1,8,64,256,512} | tee gpurun_out/serving/batcher_lanes$lanes.json
done
# Server.sign's Issuer + VerifyWithCertificate for clients outside the keyring, one per call (corpus signed on the CPU)
gcc -O2 -std=gnu99 -I include tools/serving/cert_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/cert_load
python tools/serving/make_cert_corpus.py /tmp/cert_load.bin 64 10
/tmp/cert_load /tmp/cert_load.bin 256 0 ${THREADS:-1,8,64,256,512} | tee gpurun_out/serving/cert_load.json
