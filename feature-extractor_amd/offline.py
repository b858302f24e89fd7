"""Host-side mirror of the reference's LEGACY offline analyser (struct AudioAnalyser, ref AudioAnalysis.h) on top of the
C ABI (fx_offline_*): one object stands for `num_channels` analysers.  Method names follow the reference."""
import ctypes

import numpy as np

from . import capi


class AudioAnalyser:
    def __init__(self, num_channels, nyquist_frequency=24000.0, device=0):                # ref AudioAnalysis.h:107
        self._lib = capi.load_library()
        self.num_channels = int(num_channels)
        h = ctypes.c_void_p()
        capi.check(self._lib.fx_offline_create(ctypes.byref(h), int(device), self.num_channels, float(nyquist_frequency)))
        self._h = h
        self._bins = 0          # length of previousBinMagnitudes: fixed by the first calculate_spectral_characteristics after construction / reset

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fx_offline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        capi.check(self._lib.fx_offline_reset(self._h))
        self._bins = 0

    @property
    def previous_f0(self):                                                                  # ref AudioAnalysis.h:697
        out = np.empty(self.num_channels, np.float64)
        capi.check(self._lib.fx_offline_get_previous_f0(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def _rows(self, a):
        a = np.ascontiguousarray(a, np.float32)
        if a.ndim != 2 or a.shape[0] != self.num_channels:
            raise ValueError("expected [%d][...] float32" % self.num_channels)
        return a

    def analyse_normalised_zero_crosses(self, audio, num_downsamples):                      # ref :517-541
        audio = self._rows(audio)
        out = np.empty((self.num_channels, int(num_downsamples)), np.float32)
        capi.check(self._lib.fx_offline_zero_crosses(self._h, audio.ctypes.data_as(ctypes.c_void_p), audio.shape[1], int(num_downsamples),
                                                     out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out

    def set_log_attack_time(self, energy_envelope, num_input_samples, num_downsamples, sample_rate):   # ref :611-622
        env = np.ascontiguousarray(energy_envelope, np.float32).reshape(-1)
        out = np.empty(1, np.float32)
        capi.check(self._lib.fx_offline_log_attack_time(self._h, env.ctypes.data_as(ctypes.c_void_p), env.shape[0], int(num_input_samples),
                                                        int(num_downsamples), int(sample_rate), out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out[0]

    def calculate_fft_lbp(self, fft_results, previous_fft_frame):                           # ref :543-564
        cur, prev = self._rows(fft_results), self._rows(previous_fft_frame)
        if cur.shape != prev.shape:
            raise ValueError("frames differ in shape")
        bits = np.empty(cur.shape, np.uint8)
        hi, act = np.empty(self.num_channels, np.float32), np.empty(self.num_channels, np.float32)
        capi.check(self._lib.fx_offline_fft_lbp(self._h, cur.ctypes.data_as(ctypes.c_void_p), prev.ctypes.data_as(ctypes.c_void_p), cur.shape[1],
                                                bits.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p),
                                                act.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return bits, hi, act

    def calculate_harmonic_characteristics(self, fft_results):                              # ref :253-303
        mags = self._rows(fft_results)
        out = np.empty((self.num_channels, 3), np.float32)
        capi.check(self._lib.fx_offline_harmonic_characteristics(self._h, mags.ctypes.data_as(ctypes.c_void_p), mags.shape[1],
                                                                 out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out

    def calculate_spectral_characteristics(self, fft_results):                              # ref :463-515
        """magnitudes [C][num_bins] of one frame -> [C][4] = centroid / nyquist, spread, flatness, flux; previousBinMagnitudes is kept."""
        mags = self._rows(fft_results)
        out = np.empty((self.num_channels, 4), np.float32)
        capi.check(self._lib.fx_offline_spectral_characteristics(self._h, mags.ctypes.data_as(ctypes.c_void_p), mags.shape[1],
                                                                 out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        self._bins = mags.shape[1]
        return out

    @property
    def previous_bin_magnitudes(self):                                                      # ref :700
        if self._bins == 0:         # nothing analysed since construction / reset: the reference's vector exists, zero-filled, at its window size -- which this object learns from the first frame
            return np.zeros((self.num_channels, 0), np.float64)
        out = np.empty((self.num_channels, self._bins), np.float64)
        capi.check(self._lib.fx_offline_get_previous_bins(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), self._bins))
        return out

    def calculate_normalised_spectral_slope(self, fft_results):                             # ref :566-609
        mags = self._rows(fft_results)
        out = np.empty(self.num_channels, np.float32)
        capi.check(self._lib.fx_offline_spectral_slope(self._h, mags.ctypes.data_as(ctypes.c_void_p), mags.shape[1],
                                                       out.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return out

    def analyse_auto_correlation(self, data):                                               # ref :623-665
        """data [C][num_items][2] (r, i) -> (products [C][num_items][2] of getConjugateComplexMultiplicationInPlace, peak bin [C],
        frequency [C] float64 as analyseAutoCorrelation prints it)."""
        data = np.array(data, np.float32, order="C")
        if data.ndim != 3 or data.shape[0] != self.num_channels or data.shape[2] != 2:
            raise ValueError("expected [%d][num_items][2] float32" % self.num_channels)
        peaks, freqs = np.empty(self.num_channels, np.int32), np.empty(self.num_channels, np.float64)
        capi.check(self._lib.fx_offline_auto_correlation(self._h, data.ctypes.data_as(ctypes.c_void_p), data.shape[1],
                                                         peaks.ctypes.data_as(ctypes.c_void_p), freqs.ctypes.data_as(ctypes.c_void_p), capi.MEM_HOST))
        return data, peaks, freqs
