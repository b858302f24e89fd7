"""MI355X-native per-frame audio feature path (RealTimeAnalyser -> Spectral / Harmonic / Pitch).

The product is lib/libfx_hip.so (HIP kernels behind the C ABI in include/fx.h); this package is
the thin host-side wrapper used by tests and bench.py.  There is no CPU fallback here: without
the built library and a gfx950 device, analysis calls raise.
"""
from .capi import (FEATURE_NAMES, NUM_FEATURES, FxError, load_library, library_path,
                   ONSET, RMS, F0, CENTROID, SPREAD, FLATNESS, LER, FLUX, SLOPE, HER, OER, INHARM,
                   ONSET_SPECTRAL, ONSET_AMPLITUDE, ONSET_COMBINATION,
                   ORDER_SPECTRAL_THEN_HARMONIC, ORDER_HARMONIC_THEN_SPECTRAL, ORDER_ISOLATED,
                   pack_osc12, pack_osc10, osc_encode)
from .analyser import BatchAnalyser, HopStream, PackedS24, pack_s24
from . import capi, offline, synth, wav

__all__ = ["BatchAnalyser", "HopStream", "PackedS24", "pack_s24", "FxError", "load_library", "library_path", "synth", "wav", "FEATURE_NAMES",
           "NUM_FEATURES", "pack_osc12", "pack_osc10", "osc_encode"]
