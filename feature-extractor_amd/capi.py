"""ctypes binding of include/fx.h (libfx_hip.so)."""
import ctypes
import os

import numpy as np

from . import build as _build

NUM_FEATURES = 12
(ONSET, RMS, F0, CENTROID, SPREAD, FLATNESS, LER, FLUX, SLOPE, HER, OER, INHARM) = range(12)
FEATURE_NAMES = ["onset", "rms", "f0", "centroid", "spread", "flatness", "ler", "flux",
                 "slope", "her", "oer", "inharm"]
ONSET_SPECTRAL, ONSET_AMPLITUDE, ONSET_COMBINATION = 0, 1, 2
ORDER_SPECTRAL_THEN_HARMONIC, ORDER_HARMONIC_THEN_SPECTRAL, ORDER_ISOLATED = 0, 1, 2
SPECTRAL_ONLY, HARMONIC_ONLY, LOW_LATENCY = 4, 8, 16
MEM_HOST, MEM_DEVICE = 0, 1
SAMPLE_F32, SAMPLE_F16, SAMPLE_S16, SAMPLE_S24 = 0, 1, 2, 3
FX_OK, FX_ERR_INVALID_ARGUMENT, FX_ERR_NO_DEVICE, FX_ERR_HIP, FX_ERR_OUT_OF_MEMORY, FX_ERR_UNSUPPORTED = range(6)

# every symbol include/fx.h declares
EXPORTS = ["fx_create", "fx_destroy", "fx_reset_state", "fx_set_sample_rate", "fx_set_onset_sensitivity",
           "fx_set_onset_window", "fx_set_onset_type", "fx_set_gain", "fx_push_hops", "fx_process_frames",
           "fx_push_samples", "fx_pending_samples", "fx_clear_pending", "fx_stream_submit_samples", "fx_stream_push_samples", "fx_stream_collect_samples",
           "fx_get_smoothed", "fx_sync", "fx_get_stream", "fx_last_kernel_ms", "fx_profile_begin", "fx_profile_end",
           "fx_stream_create", "fx_stream_destroy", "fx_stream_acquire", "fx_stream_submit", "fx_stream_push", "fx_stream_collect", "fx_stream_in_flight", "fx_pack_osc12",
           "fx_pack_osc10", "fx_osc_encode", "fx_last_error", "fx_abi_version",
           "fx_host_alloc", "fx_host_free", "fx_osc_message_bytes", "fx_osc_encode_batch", "fx_get_osc_datagrams", "fx_osc_sender_create", "fx_osc_sender_destroy", "fx_osc_sender_update",
           "fx_osc_sender_send", "fx_osc_sender_start", "fx_osc_sender_stop", "fx_osc_sender_get_stats", "fx_osc_receiver_create", "fx_osc_receiver_destroy",
           "fx_osc_receiver_port", "fx_osc_receiver_get_stats", "fx_osc_receiver_last",
           "fx_comm_unique_id", "fx_comm_create", "fx_comm_destroy", "fx_comm_layout", "fx_gather_smoothed", "fx_comm_sync", "fx_comm_stats",
           "fx_plan_units", "fx_twiddle_symmetry", "fx_tuning_defaults", "fx_tuning_from_env", "fx_get_tuning", "fx_set_tuning",
           "fx_offline_create", "fx_offline_destroy", "fx_offline_reset", "fx_offline_sync", "fx_offline_get_previous_f0", "fx_offline_zero_crosses",
           "fx_offline_log_attack_time", "fx_offline_fft_lbp", "fx_offline_harmonic_characteristics", "fx_offline_spectral_characteristics",
           "fx_offline_get_previous_bins", "fx_offline_spectral_slope", "fx_offline_auto_correlation"]
COMM_ID_BYTES = 128
ABI_VERSION = 6
MAX_UNITS = 24


class OscSenderStats(ctypes.Structure):
    """struct fx_osc_sender_stats of include/fx.h"""
    _fields_ = [("ticks", ctypes.c_longlong), ("late_ticks", ctypes.c_longlong), ("datagrams", ctypes.c_longlong), ("dropped", ctypes.c_longlong),
                ("syscalls", ctypes.c_longlong), ("last_tick_ms", ctypes.c_double), ("max_tick_ms", ctypes.c_double), ("total_tick_ms", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Tuning(ctypes.Structure):
    """struct fx_tuning of include/fx.h: launch-shape knobs.  Within a kernel family none changes a result bit;
    waves_per_frame selects the family (include/fx.h, FX_LOW_LATENCY) and cannot change once frames have been analysed."""
    _fields_ = [("waves_per_channel", ctypes.c_int), ("channels_per_workgroup", ctypes.c_int), ("waves_per_frame", ctypes.c_int),
                ("frames_per_unit", ctypes.c_int), ("unit_plan_len", ctypes.c_int), ("unit_plan", ctypes.c_int * MAX_UNITS),
                ("stream_graph", ctypes.c_int), ("stream_hop_kernel", ctypes.c_int), ("stream_zero_copy", ctypes.c_int),
                ("one_hop_kernel", ctypes.c_int), ("call_timing", ctypes.c_int), ("handover_spin_limit", ctypes.c_int), ("stream_fill_streaming", ctypes.c_int)]

    @classmethod
    def defaults(cls):
        t = cls()
        load_library().fx_tuning_defaults(ctypes.byref(t))
        return t

    @classmethod
    def from_env(cls):
        t = cls()
        load_library().fx_tuning_from_env(ctypes.byref(t))
        return t

    def set_plan(self, sizes):
        self.unit_plan_len = len(sizes)
        for k, v in enumerate(sizes):
            self.unit_plan[k] = int(v)
        return self


class FxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("fx error %d: %s" % (code, message))
        self.code = code


_lib = None


def library_path():
    """The shipped library -- or, for experiments only, the file FX_LIBRARY_OVERRIDE names (a variant built by tools/build_variants.py:
    selected by path, never copied over the shipped one)."""
    return os.environ.get("FX_LIBRARY_OVERRIDE") or _build.LIB_PATH


def load_library(build_if_missing=True):
    """Load libfx_hip.so; raises if it is missing and cannot be built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    override = os.environ.get("FX_LIBRARY_OVERRIDE")
    if build_if_missing and not override:
        _build.build()
    if not os.path.exists(library_path()):
        raise FxError(FX_ERR_UNSUPPORTED, "libfx_hip.so has not been built (run feature-extractor_amd/build.py)" if not override else "FX_LIBRARY_OVERRIDE names no file: %s" % override)
    # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 / librccl (same SONAMEs as
    # /opt/rocm's): if libfx_hip.so is loaded first it binds /opt/rocm's copies, a later `import torch` then brings a
    # second runtime into the process and that one finds no GPU ("No HIP GPUs are available").  Loading torch's first
    # makes both sides share one runtime, whichever order the caller imports things in.  (C++ hosts link /opt/rocm.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(library_path())
    vp, fp, i, d, f, u = ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_uint
    L.fx_create.argtypes = [ctypes.POINTER(vp), i, i, i, d, u]
    L.fx_destroy.argtypes = [vp]
    L.fx_reset_state.argtypes = [vp]
    L.fx_set_sample_rate.argtypes = [vp, d]
    L.fx_set_onset_sensitivity.argtypes = [vp, f]
    L.fx_set_onset_window.argtypes = [vp, i]
    L.fx_set_onset_type.argtypes = [vp, i]
    L.fx_set_gain.argtypes = [vp, f]
    L.fx_push_hops.argtypes = [vp, vp, i, i, i, vp, vp]
    L.fx_process_frames.argtypes = [vp, vp, i, i, i, vp, vp]
    L.fx_push_samples.argtypes = [vp, vp, i, i, i, vp, vp, ctypes.POINTER(i)]
    L.fx_pending_samples.argtypes = [vp]
    L.fx_clear_pending.argtypes = [vp]
    L.fx_stream_submit_samples.argtypes = [vp, i]
    L.fx_stream_push_samples.argtypes = [vp, vp, i, i]
    L.fx_stream_collect_samples.argtypes = [vp, vp, vp, ctypes.POINTER(i)]
    L.fx_get_smoothed.argtypes = [vp, vp, i]
    L.fx_sync.argtypes = [vp]
    L.fx_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.fx_last_kernel_ms.argtypes = [vp, fp, fp]
    L.fx_stream_create.argtypes = [vp, i, i, i, ctypes.POINTER(vp)]
    L.fx_stream_destroy.argtypes = [vp]
    L.fx_stream_acquire.argtypes = [vp, ctypes.POINTER(vp)]
    L.fx_stream_submit.argtypes = [vp]
    L.fx_stream_push.argtypes = [vp, vp, i]
    L.fx_stream_collect.argtypes = [vp, vp, vp]
    L.fx_stream_in_flight.argtypes = [vp]
    L.fx_profile_begin.argtypes = [vp]
    L.fx_profile_end.argtypes = [vp, ctypes.POINTER(d), ctypes.POINTER(d), ctypes.POINTER(i)]
    L.fx_comm_unique_id.argtypes = [vp, i]
    L.fx_comm_create.argtypes = [vp, i, i, vp, i]
    L.fx_comm_destroy.argtypes = [vp]
    L.fx_comm_layout.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i)]
    L.fx_gather_smoothed.argtypes = [vp, i, vp, i]
    L.fx_comm_sync.argtypes = [vp]
    L.fx_comm_stats.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(i), ctypes.POINTER(d), ctypes.POINTER(d)]
    L.fx_plan_units.argtypes = [i, u, i, i, ctypes.POINTER(Tuning), ctypes.POINTER(i), i]
    L.fx_twiddle_symmetry.argtypes = [i]
    L.fx_tuning_defaults.argtypes = [ctypes.POINTER(Tuning)]
    L.fx_tuning_defaults.restype = None
    L.fx_tuning_from_env.argtypes = [ctypes.POINTER(Tuning)]
    L.fx_tuning_from_env.restype = None
    L.fx_get_tuning.argtypes = [vp, ctypes.POINTER(Tuning)]
    L.fx_set_tuning.argtypes = [vp, ctypes.POINTER(Tuning)]
    L.fx_offline_create.argtypes = [ctypes.POINTER(vp), i, i, d]
    L.fx_offline_destroy.argtypes = [vp]
    L.fx_offline_reset.argtypes = [vp]
    L.fx_offline_sync.argtypes = [vp]
    L.fx_offline_get_previous_f0.argtypes = [vp, ctypes.POINTER(d)]
    L.fx_offline_zero_crosses.argtypes = [vp, vp, i, i, vp, i]
    L.fx_offline_log_attack_time.argtypes = [vp, vp, i, i, i, i, vp, i]
    L.fx_offline_fft_lbp.argtypes = [vp, vp, vp, i, vp, vp, vp, i]
    L.fx_offline_harmonic_characteristics.argtypes = [vp, vp, i, vp, i]
    L.fx_offline_spectral_characteristics.argtypes = [vp, vp, i, vp, i]
    L.fx_offline_get_previous_bins.argtypes = [vp, ctypes.POINTER(d), i]
    L.fx_offline_spectral_slope.argtypes = [vp, vp, i, vp, i]
    L.fx_offline_auto_correlation.argtypes = [vp, vp, i, vp, vp, i]
    L.fx_pack_osc12.argtypes = [fp, fp]
    L.fx_pack_osc12.restype = None
    L.fx_pack_osc10.argtypes = [fp, fp]
    L.fx_pack_osc10.restype = None
    L.fx_osc_encode.argtypes = [ctypes.c_char_p, fp, ctypes.POINTER(ctypes.c_ubyte), i]
    ip = ctypes.POINTER(i)
    ll = ctypes.c_longlong
    L.fx_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    L.fx_host_free.argtypes = [vp]
    L.fx_osc_message_bytes.argtypes = [ctypes.c_char_p, i]
    L.fx_osc_encode_batch.argtypes = [ctypes.c_char_p, i, i, fp, vp, i, ip]
    L.fx_get_osc_datagrams.argtypes = [vp, ctypes.c_char_p, i, vp, i, ip, i]
    L.fx_osc_sender_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, ctypes.c_char_p, i, u]
    L.fx_osc_sender_destroy.argtypes = [vp]
    L.fx_osc_sender_update.argtypes = [vp, vp, i, ip, i]
    L.fx_osc_sender_send.argtypes = [vp, ctypes.POINTER(ll)]
    L.fx_osc_sender_start.argtypes = [vp, d]
    L.fx_osc_sender_stop.argtypes = [vp]
    L.fx_osc_sender_get_stats.argtypes = [vp, ctypes.POINTER(OscSenderStats)]
    L.fx_osc_receiver_create.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, i, ctypes.c_char_p, i, u]
    L.fx_osc_receiver_destroy.argtypes = [vp]
    L.fx_osc_receiver_port.argtypes = [vp]
    L.fx_osc_receiver_get_stats.argtypes = [vp, ctypes.POINTER(ll), ctypes.POINTER(ll), ctypes.POINTER(ll)]
    L.fx_osc_receiver_last.argtypes = [vp, i, vp, i, ip]
    L.fx_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def check(status):
    if status != FX_OK:
        raise FxError(status, load_library().fx_last_error().decode(errors="replace"))


def plan_units(window_size, flags, waves_per_channel, num_frames, tuning=None):
    """fx_plan_units: the work-unit lengths a call of `num_frames` frames per channel is cut into (host arithmetic)."""
    buf = (ctypes.c_int * MAX_UNITS)()
    n = load_library().fx_plan_units(int(window_size), int(flags), int(waves_per_channel), int(num_frames),
                                     ctypes.byref(tuning) if tuning is not None else None, buf, MAX_UNITS)
    return list(buf[:n])


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def pack_osc12(features12):
    v = np.ascontiguousarray(features12, np.float32)
    out = np.empty(12, np.float32)
    load_library().fx_pack_osc12(_fp(v), _fp(out))
    return out


def pack_osc10(features12):
    v = np.ascontiguousarray(features12, np.float32)
    out = np.empty(10, np.float32)
    load_library().fx_pack_osc10(_fp(v), _fp(out))
    return out


def osc_encode(address, features12):
    v = np.ascontiguousarray(features12, np.float32)
    buf = (ctypes.c_ubyte * 512)()
    n = load_library().fx_osc_encode(address.encode(), _fp(v), buf, 512)
    if n < 0:
        raise FxError(FX_ERR_INVALID_ARGUMENT, "OSC address too long")
    return bytes(buf[:n])


OSC_SENDER_GSO = 1
OSC_RECEIVER_NO_GRO = 1


def osc_message_bytes(prefix, channel):
    return load_library().fx_osc_message_bytes(prefix.encode(), int(channel))


def osc_stride(prefix, first_channel, num_channels):
    """the smallest legal stride for these channels' messages: the longest message (already a multiple of 4)"""
    n = osc_message_bytes(prefix, first_channel + max(num_channels, 1) - 1)
    if n < 0:
        raise FxError(FX_ERR_INVALID_ARGUMENT, "OSC prefix too long or a negative channel number")
    return n


def osc_encode_batch(prefix, first_channel, smoothed, stride=None):
    """fx_osc_encode_batch: (datagrams uint8 [C][stride], lengths int32 [C]) for smoothed [C][12]; message c = datagrams[c, :lengths[c]]"""
    v = np.ascontiguousarray(smoothed, np.float32).reshape(-1, 12)
    stride = osc_stride(prefix, first_channel, v.shape[0]) if stride is None else int(stride)
    out = np.empty((v.shape[0], stride), np.uint8)
    lengths = np.empty(v.shape[0], np.int32)
    n = load_library().fx_osc_encode_batch(prefix.encode(), int(first_channel), v.shape[0], _fp(v), out.ctypes.data_as(ctypes.c_void_p), stride,
                                           lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if n != v.shape[0]:
        raise FxError(FX_ERR_INVALID_ARGUMENT, "fx_osc_encode_batch refused its arguments (prefix, channel range or stride)")
    return out, lengths


class OscSender:
    """fx_osc_sender: sendmmsg batches from `threads` threads to a primary and an optional secondary target, paced by start(rate_hz)."""

    def __init__(self, primary="127.0.0.1:9000", secondary=None, threads=1, gso=False):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        check(self._lib.fx_osc_sender_create(ctypes.byref(self._h), primary.encode(), secondary.encode() if secondary else None, int(threads), OSC_SENDER_GSO if gso else 0))

    def update(self, datagrams, lengths):
        d = np.ascontiguousarray(datagrams, np.uint8)
        n = np.ascontiguousarray(lengths, np.int32)
        if d.ndim != 2 or n.shape != (d.shape[0],):
            raise ValueError("datagrams [count][stride] and lengths [count]")
        check(self._lib.fx_osc_sender_update(self._h, d.ctypes.data_as(ctypes.c_void_p), d.shape[1], n.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), d.shape[0]))

    def send(self):
        sent = ctypes.c_longlong(0)
        check(self._lib.fx_osc_sender_send(self._h, ctypes.byref(sent)))
        return sent.value

    def start(self, rate_hz=60.0):
        check(self._lib.fx_osc_sender_start(self._h, float(rate_hz)))

    def stop(self):
        check(self._lib.fx_osc_sender_stop(self._h))

    def stats(self):
        st = OscSenderStats()
        check(self._lib.fx_osc_sender_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            self._lib.fx_osc_sender_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OscReceiver:
    """fx_osc_receiver: counts OSC feature messages arriving on a local UDP port (tests, benchmarks, soak runs)."""

    def __init__(self, bind="127.0.0.1:0", threads=1, prefix=None, keep_channels=0, gro=True):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        check(self._lib.fx_osc_receiver_create(ctypes.byref(self._h), bind.encode(), int(threads), prefix.encode() if prefix else None, int(keep_channels),
                                               0 if gro else OSC_RECEIVER_NO_GRO))
        self.port = self._lib.fx_osc_receiver_port(self._h)

    def stats(self):
        a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        check(self._lib.fx_osc_receiver_get_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"datagrams": a.value, "bytes": b.value, "malformed": c.value}

    def last(self, channel):
        buf = (ctypes.c_ubyte * 160)()
        n = ctypes.c_int(0)
        check(self._lib.fx_osc_receiver_last(self._h, int(channel), buf, 160, ctypes.byref(n)))
        return bytes(buf[:n.value])

    def close(self):
        if self._h:
            self._lib.fx_osc_receiver_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
