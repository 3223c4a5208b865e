"""apsu_amd — MI355X-native homomorphic query-evaluation engine for APSU (DB-side hot path).

The product is the C-ABI shared library ``libapsu_he_gpu.so`` (include/apsu_he.h), built from
``apsu_amd/csrc`` for gfx950.  This package is the thin ctypes binding used by the tests and
``bench.py``; it mirrors the names of the reference interface it replaces (seal::Evaluator as
called from receiver/apsu/receiver_osn.cpp and receiver/apsu/bin_bundle.cpp).  There is no CPU
fallback: loading fails loudly if the library is missing, and context creation fails without a GPU.
"""
from .engine import (  # noqa: F401
    ApsuHeError,
    Bundle,
    DbFile,
    HeContext,
    MultiContext,
    Powers,
    RelinKeys,
    host_alloc,
    host_free,
    lib_path,
    load_library,
    partition_bundles,
)
