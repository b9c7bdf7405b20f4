export TMPDIR=/tmp
R=$PWD
gcc -O2 -std=gnu99 -I include tools/serving/batcher_load.c -L bftkv_amd -lbftkv_gpu -lpthread -Wl,-rpath,$R/bftkv_amd -o /tmp/batcher_load
python tools/serving/make_load_corpus.py /tmp/load.bin 4096 64 > /dev/null
mkdir -p gpurun_out/serving_q
for q in 4 8; do for lanes in 3 6; do
  echo "== GPU_MAX_HW_QUEUES=$q lanes=$lanes"
  GPU_MAX_HW_QUEUES=$q /tmp/batcher_load /tmp/load.bin 256 200 $lanes 1,256,512 | tee gpurun_out/serving_q/q${q}_lanes$lanes.json | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print({k:d[k] for k in d if k in ('threads','verify_per_s','calls_per_s','p50_ms','p99_ms','wrong','sig_per_s','lone_ms','mean_ms')} or list(d)[:12])
"
done; done
