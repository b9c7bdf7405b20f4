"""bftkv_amd -- MI355X-native batched quorum verifier for yahoo/bftkv's hot path.

Package contents (only what the path needs):
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI library (include/bftkv_gpu.h)
  _native.py       ctypes binding of libbftkv_gpu.so (fails loudly when the library / GPU is absent)
  host.py          ctypes face of the host-side mirror (include/bftkv_host.h): packet framing, trust graph, wotqs,
                   vote collector, the verification sites of protocol/{client,server}.go
  dist.py          shard ranges and the verdict-bitmap layout of the exchange step (CPU / gloo tests)
  audit.py         re-verification of a storage/plain database
"""
from ._native import Batcher, Context, NativeError, load_library, LIB_PATH  # noqa: F401
