#!/bin/bash
# The HOST side of libbftkv_gpu.so under AddressSanitizer + UndefinedBehaviorSanitizer, no GPU needed: the packet / certificate /
# signature-stream parsers, the quorum mirror and the bignum set-up code read bytes a peer chose.
#   tools/sanitize_host.sh [scratch dir = /tmp/bftkv_asan] [fuzz seconds per seed = 60] [seeds = "1 2 3"]
# 1. compiles csrc/capi.hip for the host only (--offload-host-only) with -fsanitize=address,undefined (UB is fatal) and links it with
#    an empty offload bundle in place of the device code (nothing launches a kernel in the CPU suite);
# 2. copies the tree (without .git / gpurun_out) to the scratch dir, puts that library where the package loads it from, and runs
#    `pytest -m "not gpu"` there with the sanitizer runtime preloaded;
# 3. runs tools/fuzz_host_parsers.py (mutations of the certificate shapes and random bytes into every host parser) per seed;
# 4. links the same object against tools/fakehip (a stand-in HIP runtime: device memory = host memory, kernels not run) so that the
#    entry points that need a device context run too: tools/fuzz_abi_args.py (hostile counts / widths / offsets / indices / key
#    material into the C ABI, buffers sized as the arguments say) per seed, then the `-m gpu` tests themselves -- their answers are
#    wrong there (no kernels) and they FAIL; what counts is that the host side they drive prints no sanitizer report.
# Exit status 0 = the CPU suite passed and no sanitizer report was printed anywhere.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=${1:-/tmp/bftkv_asan}; SECS=${2:-60}; SEEDS=${3:-"1 2 3"}
LLVM=/opt/rocm/lib/llvm
RTD=$(dirname "$(find $LLVM/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
rm -rf "$S"; mkdir -p "$S/repo"
cd "$S"
hipcc --offload-host-only -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC \
      -c "$R/bftkv_amd/csrc/capi.hip" -o capi_host.o
SYM=$(nm -u capi_host.o | grep -o '__hip_fatbin_[0-9a-f]*' | head -1)
printf '__attribute__((aligned(4096), visibility("default"))) const char %s[4096] = "__CLANG_OFFLOAD_BUNDLE__";\n' "$SYM" > empty_bundle.c
gcc -c -fPIC empty_bundle.c -o empty_bundle.o
$LLVM/bin/clang++ -shared -fsanitize=address,undefined -shared-libsan capi_host.o empty_bundle.o -o libbftkv_gpu.so -L/opt/rocm/lib -lamdhip64 -ldl -lpthread
(cd "$R" && tar --exclude=.git --exclude=gpurun_out --exclude=__pycache__ -cf - .) | (cd "$S/repo" && tar xf -)
cp libbftkv_gpu.so "$S/repo/bftkv_amd/libbftkv_gpu.so"; touch "$S/repo/bftkv_amd/libbftkv_gpu.so"
cd "$S/repo"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_LIBRARY_PATH=$RTD LD_PRELOAD=$RTD/libclang_rt.asan-x86_64.so
python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee "$S/pytest.log" | tail -3
for seed in $SEEDS; do python tools/fuzz_host_parsers.py $seed $SECS 2>&1 | tee "$S/fuzz_$seed.log" | tail -2; done
cd "$S"
$LLVM/bin/clang++ -O1 -g -fsanitize=address,undefined -shared-libsan -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include "$R/tools/fakehip/fakehip.cpp" -o libamdhip64_fake.so
$LLVM/bin/clang++ -shared -fsanitize=address,undefined -shared-libsan capi_host.o empty_bundle.o -o libbftkv_gpu_fake.so -L. -lamdhip64_fake -ldl -lpthread -Wl,-rpath,"$S"
cp libbftkv_gpu_fake.so "$S/repo/bftkv_amd/libbftkv_gpu.so"; touch "$S/repo/bftkv_amd/libbftkv_gpu.so"
cd "$S/repo"
export LD_LIBRARY_PATH=$RTD:$S LD_PRELOAD=$RTD/libclang_rt.asan-x86_64.so:$S/libamdhip64_fake.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
for seed in $SEEDS; do python tools/fuzz_abi_args.py $seed $SECS 2>&1 | tee "$S/abi_$seed.log" | tail -c 300; echo; done
python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 --deselect tests/test_gpu_full_size.py > "$S/gpu_on_fakehip.log" 2>&1 || true
tail -1 "$S/gpu_on_fakehip.log"
if grep -l "ERROR: AddressSanitizer\|runtime error:" "$S"/*.log; then echo "sanitizer reports above"; exit 1; fi
grep -q " passed" "$S/pytest.log" && ! grep -q " failed\| error" "$S/pytest.log"
