"""heif-decoder-lib_amd — MI355X-native HEIC hot path (HEVC-intra tile reconstruction,
deblock, SAO, grid paste and fused YCbCr->RGB) behind the reference's plugin / C API.

The product is the C-ABI shared library ``libheif_mi355x.so`` (HIP kernels for gfx950 +
C++ host code, see ``include/heif_mi355x.h``).  This package is only the thin Python
harness used by tests and ``bench.py``: ctypes bindings, torch for device memory/streams.

The directory name contains hyphens, so import it with ``load_package()`` from
``__graft_entry__`` (module name ``heif_decoder_lib_amd``).
"""
from . import capi  # noqa: F401
from . import shard  # noqa: F401
from .capi import lib, HmError  # noqa: F401
