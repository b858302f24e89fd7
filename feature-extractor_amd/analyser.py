"""Host-side mirror of the reference's analyser interface on top of the C ABI.

One BatchAnalyser stands for `num_channels` AnalyserTrackControllers' analysis halves
(ref AnalyserTrackController.h:199-206): per channel a RealTimeSpectralAnalyser, a
RealTimeHarmonicAnalyser and the AudioFeatures they write.  Method names follow the reference
(RealTimeAnalyser.h:111-114, :244-258; AudioDataCollector.h:124).
"""
import ctypes

import numpy as np

from . import capi


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class PackedS24(np.ndarray):
    """uint8 bytes that ARE packed 24-bit PCM (what pack_s24 returns): the tag by which push_hops / process_frames take a byte array
    as FX_SAMPLE_S24.  A plain uint8 array is never taken for it -- 8-bit data would be read as samples three bytes wide."""


def pack_s24(values):
    """int32 samples in [-2^23, 2^23) -> packed 24-bit PCM, little endian, three bytes per sample along a new last axis folded into the
    last one: [...][n] -> PackedS24 (uint8) [...][3 n] (the layout of a 24-bit WAV file's data chunk, FX_SAMPLE_S24)."""
    v = np.ascontiguousarray(values, "<i4")
    return np.ascontiguousarray(v.view(np.uint8).reshape(v.shape + (4,))[..., :3]).reshape(v.shape[:-1] + (3 * v.shape[-1],)).view(PackedS24)


_FORMAT_NAMES = {"f32": capi.SAMPLE_F32, "f16": capi.SAMPLE_F16, "s16": capi.SAMPLE_S16, "s24": capi.SAMPLE_S24}


class BatchAnalyser:
    def __init__(self, num_channels, window_size=2048, sample_rate=48000.0, device=0,
                 order=capi.ORDER_SPECTRAL_THEN_HARMONIC, analysers="both", low_latency=False):
        """low_latency: FX_LOW_LATENCY of include/fx.h -- the kernel family for hosts that analyse one hop per call as it
        arrives and care about that hop's round trip (windows of 2048 / 4096 points: every frame on a pair of wavefronts;
        decisions identical to the default family, continuous slots may differ in the last bit)."""
        self._lib = capi.load_library()
        self.num_channels = int(num_channels)
        self.window_size = int(window_size)
        self.device = int(device)
        h = ctypes.c_void_p()
        flags = int(order) | {"both": 0, "spectral": capi.SPECTRAL_ONLY, "harmonic": capi.HARMONIC_ONLY}[analysers]
        if low_latency:
            flags |= capi.LOW_LATENCY
        capi.check(self._lib.fx_create(ctypes.byref(h), self.device, self.num_channels, self.window_size,
                                       float(sample_rate), flags))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference setters ----
    def sample_rate_changed(self, sr):                       # RealTimeAnalyser::sampleRateChanged
        capi.check(self._lib.fx_set_sample_rate(self._h, float(sr)))

    def set_onset_detection_sensitivity(self, s):            # RealTimeAnalyser.h:244
        capi.check(self._lib.fx_set_onset_sensitivity(self._h, float(s)))

    def set_onset_window_length(self, n):                    # RealTimeAnalyser.h:250
        capi.check(self._lib.fx_set_onset_window(self._h, int(n)))

    def set_onset_detection_type(self, t):                   # RealTimeAnalyser.h:258
        capi.check(self._lib.fx_set_onset_type(self._h, int(t)))

    def set_gain(self, g):                                   # AudioDataCollector::setGain
        capi.check(self._lib.fx_set_gain(self._h, float(g)))

    def reset_state(self):
        capi.check(self._lib.fx_reset_state(self._h))

    # ---- launch-shape knobs (within a kernel family they never change a result bit; waves_per_frame selects the family) ----
    def get_tuning(self):
        t = capi.Tuning()
        capi.check(self._lib.fx_get_tuning(self._h, ctypes.byref(t)))
        return t

    def set_tuning(self, tuning=None, **knobs):
        """Replace the context's knobs (struct fx_tuning); keyword arguments change single fields of the current ones."""
        t = tuning if tuning is not None else self.get_tuning()
        names = {f[0] for f in capi.Tuning._fields_}
        for k, v in knobs.items():
            if k not in names:                      # (setattr on a ctypes.Structure would take a mistyped name silently)
                raise ValueError("fx_tuning has no knob %r (it has: %s)" % (k, ", ".join(sorted(names))))
            if k == "unit_plan":
                t.set_plan(v)
            else:
                setattr(t, k, int(v))
        capi.check(self._lib.fx_set_tuning(self._h, ctypes.byref(t)))
        return t

    def set_test_hooks(self, bits):
        """fx_set_tuning_internal (csrc/fx_kernels.h, FX_HOOK_*): tests only, not part of include/fx.h."""
        fn = self._lib.fx_set_tuning_internal
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_uint], ctypes.c_int
        capi.check(fn(self._h, int(bits)))

    def sync(self):
        capi.check(self._lib.fx_sync(self._h))

    def last_kernel_ms(self):
        a, b = ctypes.c_float(), ctypes.c_float()
        capi.check(self._lib.fx_last_kernel_ms(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def profile_begin(self):
        capi.check(self._lib.fx_profile_begin(self._h))

    def profile_end(self):
        """(frame kernel ms, smoothing/onset kernels ms, calls) summed since profile_begin()."""
        a, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        capi.check(self._lib.fx_profile_end(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return a.value, b.value, n.value

    def stream(self):
        s = ctypes.c_void_p()
        capi.check(self._lib.fx_get_stream(self._h, ctypes.byref(s)))
        return s.value

    # ---- analysis ----
    def _run(self, fn, x, per_frame, want_raw, want_smoothed, out_raw=None, out_smoothed=None, sample_format=None):
        """Sample formats by dtype: float32, float16, int16 (16-bit PCM); packed 24-bit PCM (three bytes per sample) is a PackedS24
        array (pack_s24) or any uint8 buffer passed with sample_format="s24" -- never inferred from dtype uint8 alone.  The integer
        formats are widened in the kernels' load stage to exactly the floats a WAV reader would produce.  Device buffers must start
        on a 16-byte boundary (fx.h)."""
        C = self.num_channels
        if sample_format is not None and sample_format not in _FORMAT_NAMES:
            raise ValueError("sample_format must be one of %s" % ", ".join(sorted(_FORMAT_NAMES)))
        want = None if sample_format is None else _FORMAT_NAMES[sample_format]
        s24_error = 'uint8 samples are taken as packed 24-bit PCM only with sample_format="s24" (or as a PackedS24 array from pack_s24)'
        if _is_torch(x):
            import torch
            if not x.is_cuda:
                raise ValueError("torch input must live on the GPU (use numpy for host buffers)")
            if not x.is_contiguous():
                raise ValueError("device input must be contiguous")
            if x.dtype == torch.float32:
                fmt = capi.SAMPLE_F32
            elif x.dtype == torch.float16:
                fmt = capi.SAMPLE_F16
            elif x.dtype == torch.int16:
                fmt = capi.SAMPLE_S16            # 16-bit PCM: v / 32768 in the kernels' load stage (include/fx_wav.hpp's scaling)
            elif x.dtype == torch.uint8:
                if want != capi.SAMPLE_S24:
                    raise ValueError(s24_error)
                fmt = capi.SAMPLE_S24            # packed 24-bit PCM: three bytes per sample
                per_frame = 3 * per_frame
            else:
                raise ValueError("samples must be float32, float16, int16 (16-bit PCM) or uint8 (packed 24-bit PCM)")
            if want is not None and want != fmt:
                raise ValueError("sample_format=%r does not describe a %s tensor" % (sample_format, x.dtype))
            if x.data_ptr() % 16:
                raise ValueError("device input must start on a 16-byte boundary (an offset view of a tensor may not)")
            if x.numel() % (C * per_frame):
                raise ValueError("input size is not a multiple of channels x samples per frame")
            T = x.numel() // (C * per_frame)
            if x.device.index != self.device:
                raise ValueError("input lives on %s, the analyser on cuda:%d" % (x.device, self.device))
            for name, o in (("out_raw", out_raw), ("out_smoothed", out_smoothed)):
                if o is not None:
                    self._check_out(o, C * T * 12, name)
            raw = out_raw if out_raw is not None else (torch.empty((C, T, 12), dtype=torch.float32, device=x.device) if want_raw else None)
            sm = out_smoothed if out_smoothed is not None else (torch.empty((C, T, 12), dtype=torch.float32, device=x.device) if want_smoothed else None)
            # The library enqueues on its own HIP stream.  Order it after the producer of x on torch's current
            # stream, and torch's current stream after the analysis, both on the device (no host sync).  Because
            # the current stream waits for the analysis, anything torch later does with these blocks on that
            # stream -- reading the results, freeing and recycling x -- is ordered after the kernels that use
            # them.  (Tensor.record_stream is deliberately not used: the allocator would record events on the
            # library's stream when the tensors die, possibly after fx_destroy has destroyed that stream.)
            # A caller that works ON the library's stream (`with torch.cuda.stream(analyser.torch_stream()):`) needs no ordering
            # at all -- and the two waits are half of a one-hop call's cost from Python (25 of 56 us: tools/py_call_overhead.py).
            cur = torch.cuda.current_stream(x.device)
            lib = self._torch_stream(x.device)
            foreign = cur.cuda_stream != lib.cuda_stream
            if foreign:
                lib.wait_stream(cur)
            capi.check(fn(self._h, ctypes.c_void_p(x.data_ptr()), T, fmt, capi.MEM_DEVICE,
                          ctypes.c_void_p(raw.data_ptr()) if raw is not None else None,
                          ctypes.c_void_p(sm.data_ptr()) if sm is not None else None))
            if foreign:
                cur.wait_stream(lib)
            return raw, sm
        tagged = isinstance(x, PackedS24)
        x = np.ascontiguousarray(x)
        if x.dtype == np.float16:
            fmt = capi.SAMPLE_F16
        elif x.dtype == np.int16:
            fmt = capi.SAMPLE_S16
        elif x.dtype == np.uint8:
            if not (tagged or want == capi.SAMPLE_S24):
                raise ValueError(s24_error)
            fmt = capi.SAMPLE_S24
            per_frame = 3 * per_frame
        else:
            x = np.ascontiguousarray(x, np.float32)
            fmt = capi.SAMPLE_F32
        if want is not None and want != fmt:
            raise ValueError("sample_format=%r does not describe a %s array" % (sample_format, x.dtype))
        if x.size % (C * per_frame):
            raise ValueError("input size is not a multiple of channels x samples per frame")
        T = x.size // (C * per_frame)
        raw = np.empty((C, T, 12), np.float32) if want_raw else None
        sm = np.empty((C, T, 12), np.float32) if want_smoothed else None
        capi.check(fn(self._h, x.ctypes.data_as(ctypes.c_void_p), T, fmt, capi.MEM_HOST,
                      raw.ctypes.data_as(ctypes.c_void_p) if raw is not None else None,
                      sm.ctypes.data_as(ctypes.c_void_p) if sm is not None else None))
        return raw, sm

    def torch_stream(self):
        """The library's stream as a torch stream.  Inside `with torch.cuda.stream(analyser.torch_stream()):` the producer of
        the samples, the analysis and the consumer of the results are ordered by the stream itself, and device-buffer calls
        skip the cross-stream waits they otherwise make against torch's current stream."""
        import torch
        return self._torch_stream(torch.device("cuda", self.device))

    def _torch_stream(self, device):
        """The library's hipStream_t as a torch stream (for device-side ordering against torch's streams)."""
        import torch
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.stream(), device=device)
        return self._ext_stream

    def _check_out(self, o, numel, name):
        if not (_is_torch(o) and o.is_cuda and o.device.index == self.device and o.is_contiguous()
                and str(o.dtype) == "torch.float32" and o.numel() == numel):
            raise ValueError("%s must be a contiguous float32 CUDA tensor on cuda:%d with %d elements" % (name, self.device, numel))

    def push_hops(self, hops, want_raw=True, want_smoothed=True, out_raw=None, out_smoothed=None, sample_format=None):
        """hops [C][T][N/2] -> (raw [C][T][12], smoothed [C][T][12])."""
        return self._run(self._lib.fx_push_hops, hops, self.window_size // 2, want_raw, want_smoothed, out_raw, out_smoothed, sample_format)

    def process_frames(self, frames, want_raw=True, want_smoothed=True, out_raw=None, out_smoothed=None, sample_format=None):
        """frames [C][T][N] -> (raw [C][T][12], smoothed [C][T][12])."""
        return self._run(self._lib.fx_process_frames, frames, self.window_size, want_raw, want_smoothed, out_raw, out_smoothed, sample_format)

    # ---- the collector's interface: device blocks of any length (ref AudioDataCollector.h:36-94) ----
    def pending_samples(self):
        """samples per channel that fx_push_samples is holding back (< window_size / 2)"""
        return int(self._lib.fx_pending_samples(self._h))

    def clear_buffer(self):                                  # AudioDataCollector::clearBuffer, AudioDataCollector.h:122
        capi.check(self._lib.fx_clear_pending(self._h))

    def push_samples(self, samples, want_raw=True, want_smoothed=True, sample_format=None):
        """samples [C][n] for ANY n >= 0 (a device block: 441, 480, 512 ... samples per channel) -> (raw [C][frames][12], smoothed
        [C][frames][12]) with frames = (pending + n) // (window_size / 2); what is left over stays pending in device memory.  Same bits
        as push_hops on the same stream cut into hops.  numpy (host) or torch CUDA tensors, formats as push_hops."""
        C, H = self.num_channels, self.window_size // 2
        if sample_format is not None and sample_format not in _FORMAT_NAMES:
            raise ValueError("sample_format must be one of %s" % ", ".join(sorted(_FORMAT_NAMES)))
        want = None if sample_format is None else _FORMAT_NAMES[sample_format]
        frames_out = ctypes.c_int(0)
        if _is_torch(samples):
            import torch
            x = samples
            if not x.is_cuda or not x.is_contiguous() or x.device.index != self.device:
                raise ValueError("torch input must be a contiguous tensor on cuda:%d" % self.device)
            fmt = {torch.float32: capi.SAMPLE_F32, torch.float16: capi.SAMPLE_F16, torch.int16: capi.SAMPLE_S16, torch.uint8: capi.SAMPLE_S24}.get(x.dtype)
            if fmt is None or (fmt == capi.SAMPLE_S24 and want != capi.SAMPLE_S24) or (want is not None and want != fmt):
                raise ValueError("samples must be float32, float16, int16, or uint8 with sample_format=\"s24\"")
            per = 3 if fmt == capi.SAMPLE_S24 else 1
            if x.numel() % (C * per):
                raise ValueError("input size is not a multiple of the channel count")
            n = x.numel() // (C * per)
            if x.data_ptr() % 4:
                raise ValueError("device input must start on a 4-byte boundary")
            frames = (self.pending_samples() + n) // H
            raw = torch.empty((C, frames, 12), dtype=torch.float32, device=x.device) if want_raw else None
            sm = torch.empty((C, frames, 12), dtype=torch.float32, device=x.device) if want_smoothed else None
            cur = torch.cuda.current_stream(x.device)
            lib = self._torch_stream(x.device)
            foreign = cur.cuda_stream != lib.cuda_stream
            if foreign:
                lib.wait_stream(cur)
            capi.check(self._lib.fx_push_samples(self._h, ctypes.c_void_p(x.data_ptr()), n, fmt, capi.MEM_DEVICE,
                                                 ctypes.c_void_p(raw.data_ptr()) if raw is not None and frames else None,
                                                 ctypes.c_void_p(sm.data_ptr()) if sm is not None and frames else None, ctypes.byref(frames_out)))
            if foreign:
                cur.wait_stream(lib)
            assert frames_out.value == frames
            return raw, sm
        tagged = isinstance(samples, PackedS24)
        x = np.ascontiguousarray(samples)
        if x.dtype == np.float16:
            fmt = capi.SAMPLE_F16
        elif x.dtype == np.int16:
            fmt = capi.SAMPLE_S16
        elif x.dtype == np.uint8:
            if not (tagged or want == capi.SAMPLE_S24):
                raise ValueError('uint8 samples are packed 24-bit PCM only with sample_format="s24" (or as a PackedS24 array)')
            fmt = capi.SAMPLE_S24
        else:
            x = np.ascontiguousarray(x, np.float32)
            fmt = capi.SAMPLE_F32
        if want is not None and want != fmt:
            raise ValueError("sample_format=%r does not describe a %s array" % (sample_format, x.dtype))
        per = 3 if fmt == capi.SAMPLE_S24 else 1
        if x.size % (C * per):
            raise ValueError("input size is not a multiple of the channel count")
        n = x.size // (C * per)
        frames = (self.pending_samples() + n) // H
        raw = np.empty((C, frames, 12), np.float32) if want_raw else None
        sm = np.empty((C, frames, 12), np.float32) if want_smoothed else None
        capi.check(self._lib.fx_push_samples(self._h, x.ctypes.data_as(ctypes.c_void_p), n, fmt, capi.MEM_HOST,
                                             raw.ctypes.data_as(ctypes.c_void_p) if raw is not None and frames else None,
                                             sm.ctypes.data_as(ctypes.c_void_p) if sm is not None and frames else None, ctypes.byref(frames_out)))
        assert frames_out.value == frames
        return raw, sm

    def get_features(self, out=None):
        """Latest AudioFeatures::getValue of every slot, [C][12]: a host array, or -- with `out`, a contiguous
        float32 CUDA tensor of that shape -- an asynchronous device copy on the library's stream (what the OSC
        sink of a sharded run gathers)."""
        if out is not None:
            import torch
            self._check_out(out, self.num_channels * 12, "out")
            cur = torch.cuda.current_stream(out.device)
            lib = self._torch_stream(out.device)
            lib.wait_stream(cur)
            capi.check(self._lib.fx_get_smoothed(self._h, ctypes.c_void_p(out.data_ptr()), capi.MEM_DEVICE))
            cur.wait_stream(lib)
            return out
        out = np.empty((self.num_channels, 12), np.float32)
        capi.check(self._lib.fx_get_smoothed(self._h, out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out

    def osc_datagrams(self, prefix="/Audio/A", first_channel=0, stride=None):
        """fx_get_osc_datagrams: every channel's wire-ready OSC feature message, written on the device from the latest smoothed vectors
        (ref OSCFeatureAnalysisOutput.h:89-113, MainComponent.cpp:170).  Returns (datagrams uint8 [C][stride], lengths int32 [C])."""
        stride = capi.osc_stride(prefix, first_channel, self.num_channels) if stride is None else int(stride)
        out = np.empty((self.num_channels, stride), np.uint8)
        lengths = np.empty(self.num_channels, np.int32)
        capi.check(self._lib.fx_get_osc_datagrams(self._h, prefix.encode(), int(first_channel), out.ctypes.data_as(ctypes.c_void_p), stride,
                                                  lengths.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), capi.MEM_HOST))
        return out, lengths

    # ---- multi-GPU: gather of the latest smoothed vectors to the OSC sink rank (RCCL, through the C ABI) ----
    @staticmethod
    def comm_unique_id():
        """bytes to hand to every rank's comm_create (rank 0 creates them)."""
        buf = ctypes.create_string_buffer(capi.COMM_ID_BYTES)
        capi.check(capi.load_library().fx_comm_unique_id(buf, capi.COMM_ID_BYTES))
        return buf.raw

    def comm_create(self, rank, world_size, unique_id):
        capi.check(self._lib.fx_comm_create(self._h, int(rank), int(world_size), ctypes.c_char_p(unique_id), len(unique_id)))
        self._world = int(world_size)

    def comm_destroy(self):
        capi.check(self._lib.fx_comm_destroy(self._h))

    def comm_layout(self):
        """(total channels over all ranks, [first channel of rank r])."""
        total = ctypes.c_int()
        first = (ctypes.c_int * self._world)()
        capi.check(self._lib.fx_comm_layout(self._h, ctypes.byref(total), first))
        return total.value, list(first)

    def gather_features(self, dst=0, out=None):
        """Asynchronous gather of every rank's latest smoothed vectors to rank `dst`.  `out`: on dst a host
        array or contiguous float32 CUDA tensor of [total_channels][12]; valid after comm_sync()."""
        if out is None:
            capi.check(self._lib.fx_gather_smoothed(self._h, int(dst), None, capi.MEM_DEVICE))
        elif _is_torch(out):
            total, _ = self.comm_layout()
            self._check_out(out, total * 12, "out")
            capi.check(self._lib.fx_gather_smoothed(self._h, int(dst), ctypes.c_void_p(out.data_ptr()), capi.MEM_DEVICE))
        else:
            if not (out.dtype == np.float32 and out.flags.c_contiguous):
                raise ValueError("out must be a C-contiguous float32 array")
            capi.check(self._lib.fx_gather_smoothed(self._h, int(dst), out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out

    def comm_sync(self):
        capi.check(self._lib.fx_comm_sync(self._h))

    def comm_stats(self):
        """fx_comm_stats: RCCL's rank count, gathers issued / timed, their summed and longest device time (ms) on the side stream."""
        n, g, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        tot, mx = ctypes.c_double(), ctypes.c_double()
        capi.check(self._lib.fx_comm_stats(self._h, ctypes.byref(n), ctypes.byref(g), ctypes.byref(t), ctypes.byref(tot), ctypes.byref(mx)))
        return {"rccl_ranks": n.value, "gathers": g.value, "gathers_timed": t.value, "gather_ms_total": tot.value, "gather_ms_max": mx.value,
                "gather_ms_mean": tot.value / t.value if t.value else None}


class HopStream:
    """Streaming ingest on top of a BatchAnalyser: the stand-in for AudioDataCollector's ring
    (ref AudioDataCollector.h:36-94).  Batches of `hops_per_batch` hops per channel are written into
    pinned host slots; the copy to the GPU runs on a side stream and overlaps the previous batch's
    analysis; results come back in order."""

    def __init__(self, analyser, hops_per_batch, slots=3, dtype=np.float32):
        self._an = analyser
        self._lib = analyser._lib
        self.hops = int(hops_per_batch)
        self.slots = int(slots)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float16), np.dtype(np.int16), np.dtype(np.uint8)):
            raise ValueError("HopStream samples are float32, float16, int16 (16-bit PCM) or uint8 (packed 24-bit PCM, three bytes per sample)")
        fmt = {np.dtype(np.float16): capi.SAMPLE_F16, np.dtype(np.int16): capi.SAMPLE_S16, np.dtype(np.uint8): capi.SAMPLE_S24}.get(self.dtype, capi.SAMPLE_F32)
        h = ctypes.c_void_p()
        capi.check(self._lib.fx_stream_create(analyser._h, self.hops, self.slots, fmt, ctypes.byref(h)))
        self._h = h
        self._shape = (analyser.num_channels, self.hops, (analyser.window_size // 2) * (3 if self.dtype == np.uint8 else 1))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fx_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def in_flight(self):
        return int(self._lib.fx_stream_in_flight(self._h))

    def slot(self):
        """numpy view [C][hops][N/2] of the next pinned slot to fill."""
        p = ctypes.c_void_p()
        capi.check(self._lib.fx_stream_acquire(self._h, ctypes.byref(p)))
        n = int(np.prod(self._shape))
        buf = (ctypes.c_char * (n * self.dtype.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=self.dtype).reshape(self._shape)

    def submit(self):
        capi.check(self._lib.fx_stream_submit(self._h))

    def push(self, hops, fill_threads=1):
        """Copy one batch into the next slot and submit it.  fill_threads > 1: the copy is made by that many host threads inside
        the library (fx_stream_push) -- for batches of many megabytes, where one thread's memcpy is several times slower than PCIe."""
        hops = np.asarray(hops, self.dtype)
        if hops.size != int(np.prod(self._shape)):
            raise ValueError("a batch is %r samples" % (self._shape,))
        if fill_threads > 1:
            hops = np.ascontiguousarray(hops)
            capi.check(self._lib.fx_stream_push(self._h, hops.ctypes.data_as(ctypes.c_void_p), int(fill_threads)))
            return
        self.slot()[...] = hops.reshape(self._shape)
        self.submit()

    def push_samples(self, samples, fill_threads=1):
        """A block of n samples per channel, [C][n] with 0 <= n <= hops_per_batch * window_size / 2 (a device callback's block), into the
        next slot: fx_stream_push_samples.  collect_samples() returns what the context's pending samples and the block yielded.
        (No sample_format argument, unlike BatchAnalyser.push_samples: a ring's format is fixed when it is created -- `dtype` of HopStream,
        uint8 = packed 24-bit -- and the block is converted to it.)"""
        x = np.ascontiguousarray(samples, self.dtype)
        per = 3 if self.dtype == np.uint8 else 1
        C = self._shape[0]
        if x.size % (C * per):
            raise ValueError("input size is not a multiple of the channel count")
        capi.check(self._lib.fx_stream_push_samples(self._h, x.ctypes.data_as(ctypes.c_void_p), x.size // (C * per), int(fill_threads)))

    def collect_samples(self, want_raw=True, want_smoothed=True):
        """(raw [C][frames][12], smoothed [C][frames][12]) of the oldest batch; frames may be 0 for a block that completed no hop."""
        C = self._shape[0]
        raw = np.empty((C, self.hops, 12), np.float32)
        sm = np.empty((C, self.hops, 12), np.float32)
        n = ctypes.c_int(0)
        capi.check(self._lib.fx_stream_collect_samples(self._h, raw.ctypes.data_as(ctypes.c_void_p), sm.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)))
        f = n.value
        flat_r, flat_s = raw.reshape(-1)[:C * f * 12].reshape(C, f, 12), sm.reshape(-1)[:C * f * 12].reshape(C, f, 12)
        return (flat_r.copy() if want_raw else None), (flat_s.copy() if want_smoothed else None)

    def collect(self, want_raw=True, want_smoothed=True):
        C = self._shape[0]
        raw = np.empty((C, self.hops, 12), np.float32) if want_raw else None
        sm = np.empty((C, self.hops, 12), np.float32) if want_smoothed else None
        capi.check(self._lib.fx_stream_collect(self._h,
                                               raw.ctypes.data_as(ctypes.c_void_p) if raw is not None else None,
                                               sm.ctypes.data_as(ctypes.c_void_p) if sm is not None else None))
        return raw, sm
