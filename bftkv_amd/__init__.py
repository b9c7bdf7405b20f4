"""bftkv_amd -- MI355X-native batched quorum verifier for yahoo/bftkv's hot path.

Package contents (only what the path needs):
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI library (include/bftkv_gpu.h)
  _native.py       ctypes binding of libbftkv_gpu.so (fails loudly when the library / GPU is absent)
  crypto_gpu.py    host-side mirror of crypto.Signature / crypto.CollectiveSignature
                   (crypto/crypto.go:50-71) over the C ABI
"""
from ._native import Batcher, Context, NativeError, load_library, LIB_PATH  # noqa: F401
