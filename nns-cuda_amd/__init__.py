"""nns-cuda_amd — MI355X-native brute-force nearest-neighbour hot path.

Contents: csrc/ (hand-written HIP for gfx950 behind the C ABI of include/nns.h,
built into libnns_mi355x.so) and host.py (ctypes mirror of the reference's
``cudaCall`` entry point plus the device-resident split API).  The directory
name is not a Python identifier; load it with ``__graft_entry__.load_package()``.
"""
from .host import *  # noqa: F401,F403
from .host import (ABI_SYMBOLS, LIB_PATH, NNS_KEY_NONE, Index, NNSError, allreduce_min_keys,  # noqa: F401
                   cudaCall, device_count, fill_uniform, keys_min, keys_unpack, lib, search, search_bf16, search_multi, selftest_mfma, shard_range, trim, warmup,
                   to_bf16_bits)
