#!/bin/bash
# The HOST side of libbftkv_gpu.so under ThreadSanitizer, no GPU needed: the micro-batcher (door, leaders, lanes), the pipelined
# host-buffer call's helper threads, forked contexts against a writer replacing the key table -- the code whose failures are races,
# which the GPU suite can only meet by luck.
#   tools/tsan_host.sh [scratch dir = /tmp/bftkv_tsan] [seconds per run = 20]
# 1. tools/fakehip/fakehip.cpp stands in for libamdhip64 (device memory = host memory, synchronous streams, kernels not run, only
#    k_finish_staged's publication of a staged call's results emulated: the host waits for it without a synchronisation);
# 2. csrc/capi.hip compiled for the host only with -fsanitize=thread and linked against it;
# 3. tools/serving/batcher_load.c (1..256 callers of one entry) and tools/fakehip/stress.c (every batcher entry, host-buffer calls on
#    forks and a keyring / quorum writer at once) run on a corpus whose signatures nobody checks.
# Exit status 0 = every call returned, nothing failed open, and TSan printed no report.  BFTKV_TSAN_QUICK=1: the mixed stress run
# alone (tests/test_host_concurrency.py runs that form in the CPU suite).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=${1:-/tmp/bftkv_tsan}; SECS=${2:-20}
LLVM=/opt/rocm/lib/llvm
RTD=$(dirname "$(find $LLVM/lib/clang -name 'libclang_rt.tsan-x86_64.so' | head -1)")
SAN="-O1 -g -fsanitize=thread -fno-omit-frame-pointer"
mkdir -p "$S"; cd "$S"
$LLVM/bin/clang++ $SAN -shared-libsan -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$R/tools/fakehip/fakehip.cpp" -o libamdhip64_fake.so
hipcc --offload-host-only $SAN -std=c++17 -fPIC -c "$R/bftkv_amd/csrc/capi.hip" -o capi_host.o 2> hipcc.log || { cat hipcc.log; exit 1; }
SYM=$(nm -u capi_host.o | grep -o '__hip_fatbin_[0-9a-f]*' | head -1)
printf '__attribute__((aligned(4096), visibility("default"))) const char %s[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n' "$SYM" > empty_bundle.c
gcc -c -fPIC empty_bundle.c -o empty_bundle.o
$LLVM/bin/clang++ -shared -fsanitize=thread -shared-libsan capi_host.o empty_bundle.o -o libbftkv_gpu.so -L. -lamdhip64_fake -ldl -lpthread -Wl,-rpath,"$S"
for t in serving/batcher_load fakehip/stress; do
  $LLVM/bin/clang $SAN -shared-libsan -std=gnu11 -I "$R/include" "$R/tools/$t.c" -L. -lbftkv_gpu -lpthread -Wl,-rpath,"$S" -o "$(basename $t)"
done
# cfg-2 shaped writes whose "signatures" are the encoded messages themselves: nothing here verifies them
python - "$R" "$S/load.bin" <<'PY'
import struct, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from corpus import build as cb
reps, n = 64, 512
cl = cb.make_cluster(reps)
c = cb.make_write_corpus(cl, n, batch_signer=lambda em, ki: em)
f, mn, thr, suff = cb.quorum_numbers(reps)
with open(sys.argv[2], "wb") as fh:
    fh.write(struct.pack("<I", reps))
    for r in cl.replicas:
        fh.write(struct.pack("<Q", r.key_id) + r.n.to_bytes(256, "big") + r.e.to_bytes(4, "big"))
    fh.write(struct.pack("<iiiiI", f, mn, thr, suff, n))
    fh.write(c.tbss_off.astype("<u8").tobytes() + c.ss_off.astype("<u8").tobytes())
    fh.write(c.tbss_blob.tobytes() + c.ss_blob.tobytes())
    fh.write(((c.expected_valid >= cl.suff).astype(np.uint8)).tobytes())
PY
python "$R/tools/fakehip/make_extras.py" "$S/extras.bin"
export LD_LIBRARY_PATH=$RTD:$S TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1"
CALLERS=8; PAUSE=1000
if [ -n "$BFTKV_TSAN_QUICK" ]; then CALLERS=4; PAUSE=400; fi
: > load.err; : > stress_1lane.err; : > stress_1lane.out
if [ -z "$BFTKV_TSAN_QUICK" ]; then
  ./batcher_load load.bin 256 200 4 1,8,64,256 > load.out 2> load.err || true
  ./stress load.bin extras.bin "$SECS" 8 1 0 > stress_1lane.out 2> stress_1lane.err || { echo "stress (one lane): failed"; cat stress_1lane.out; exit 1; }
fi
./stress load.bin extras.bin "$SECS" $CALLERS 3 $PAUSE > stress.out 2> stress.err || { echo "stress: a call failed open or did not return"; cat stress.out; exit 1; }
cat stress.out stress_1lane.out
REPORTS=$(grep -h "SUMMARY: ThreadSanitizer" load.err stress.err stress_1lane.err | sort | uniq -c || true)
if [ -n "$REPORTS" ]; then echo "$REPORTS"; echo "ThreadSanitizer reports above (full text in $S/*.err)"; exit 1; fi
echo "ThreadSanitizer: no report"
