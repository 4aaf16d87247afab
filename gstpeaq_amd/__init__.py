"""gstpeaq_amd -- MI355X-native PEAQ (ITU-R BS.1387) engine behind the plugin
surface of HSU-ANT/gstpeaq.

The product is the C-ABI shared library ``libpeaq_amd.so`` (include/peaq_amd.h)
built from gstpeaq_amd/csrc (hand-written HIP for gfx950).  This Python package
is only the thin ctypes binding used by the tests and by bench.py; PyTorch is
used for device memory and streams, nothing else.  There is no CPU fallback:
every entry point raises if the HIP library or the GPU is missing.
"""
from .capi import (Broker, Context, PeaqError, Session, batch_run, build_library, debug_backend, debug_frontend, debug_filterbank, run_pair,  # noqa: F401
                   library_path, load_library, synth_fill, MOV_NAMES_BASIC, MOV_NAMES_ADVANCED)

__all__ = ["Broker", "Context", "PeaqError", "Session", "batch_run", "build_library", "debug_backend", "debug_frontend", "debug_filterbank", "run_pair",
           "library_path", "load_library", "synth_fill", "MOV_NAMES_BASIC", "MOV_NAMES_ADVANCED"]
