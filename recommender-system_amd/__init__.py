"""MI355X-native matrix-factorisation hot path (drop-in for vladstojna/recommender-system's matFact).

The product is the C ABI in include/matfact_hip.h (csrc/libmatfact_hip.so, hand-written gfx950 kernels) and
the C host pieces in include/matfact_host.h (host/libmatfact_host.so, host/matFact).  This Python package is
only the binding layer used by tests/, bench.py and the one-process-per-GPU driver (`sharded.py`); it holds
no compute of its own and raises if the HIP library is missing -- there is no CPU fallback.
"""
from . import capi, sharded  # noqa: F401
from .capi import Plan, HipBackendError  # noqa: F401
