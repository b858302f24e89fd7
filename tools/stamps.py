"""Diagnostic: build a -DFX_STAMPS variant of the library into gpurun_out/, run the bench shape once and
print each section's share of the wave cycles.  Shares only -- a stamped build is never timed."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libfx_hip_stamps.so")
b = importlib.import_module("feature-extractor_amd.build")
if not os.path.exists(out) or "--rebuild" in sys.argv:
    objs = []
    for src in b.SOURCES:
        o = os.path.join(ROOT, "gpurun_out", os.path.splitext(src)[0] + "_st.o")
        subprocess.check_call([b._hipcc()] + b.HIPCC_FLAGS + ["-DFX_STAMPS"] + os.environ.get("FX_EXTRA_FLAGS", "").split() + ["-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
if "--build-only" in sys.argv:
    sys.exit(0)
b.LIB_PATH = out
fx = importlib.import_module("feature-extractor_amd")
import torch
N = int(os.environ.get("N", 1024)); C = int(os.environ.get("C", 1024)); T = int(os.environ.get("T", 64))
an = fx.BatchAnalyser(C, N)
x = torch.from_numpy(fx.synth.frames(C, T, N)).cuda()
for _ in range(3):
    an.process_frames(x, want_raw=False)
an.sync()
lib = fx.load_library()
st = (ctypes.c_ulonglong * 16)()
lib.fx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.fx_debug_stamps(an._h, st, 1)
an.process_frames(x, want_raw=False); an.sync()
kernel_ms = an.last_kernel_ms()[0]
lib.fx_debug_stamps(an._h, st, 0)
names = ["load", "rms", "spectral fft", "spectral sums/flux/flatness", "raw fft + harmonic sums", "low-pass", "pitch fft",
         "power + inverse fft", "v + lag scan", "harmonic part 2", "store", "loop overhead",
         "  load: address setup", "  load: window load+LDS write", "(unused)", "(unused)"]
tot = float(sum(st))
for n, v in zip(names, st):
    print("%-30s %6.2f %%   %8.0f cycles/frame" % (n, 100 * v / tot, v / (C * T)))
print("total %.0f ticks/frame/wave" % (tot / (C * T)))
import math
waves = 8 if T >= 8 else T
frames_per_wave = math.ceil(T / waves)
print("stamped kernel %.3f ms; %d frames per wave -> %.1f us per frame per wave -> %.2f ticks/ns" % (
    kernel_ms, frames_per_wave, kernel_ms * 1e3 / frames_per_wave, (tot / (C * T)) / (kernel_ms * 1e6 / frames_per_wave)))
