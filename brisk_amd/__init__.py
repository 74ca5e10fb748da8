"""brisk_amd -- MI355X-native Brisk hot path.

The product is ``libbrisk_hip.so`` (HIP kernels + the C-ABI of
``include/brisk_hip.h``) and the C++ facade in ``brisk_amd/include``.  This
Python package is a thin ctypes view of the C-ABI used by tests, ``bench.py``
and ``__graft_entry__``; it never computes anything itself and raises if the
library is missing (there is no CPU fallback).
"""
from .hipapi import BriskHip, BriskHipError, build_apps, build_library, library_path, coef_table  # noqa: F401
