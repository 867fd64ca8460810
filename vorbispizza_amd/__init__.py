"""vorbispizza_amd -- MI355X-native Vorbis PCM-synthesis back end (host-side Python binding).

The product is the C-ABI library built from vorbispizza_amd/csrc (include/vorbispizza_synth.h);
this package is the ctypes view of it used by tests and bench.py.  No CPU fallback exists.
"""
from . import capi  # noqa: F401
from .capi import Context, Decoder, SynthError, make_packets  # noqa: F401

__all__ = ["capi", "Context", "Decoder", "SynthError", "make_packets"]
