"""Randomised parity stress: HIP path (through the C ABI) vs the CPU oracle over random signal mixes,
levels, window sizes, batch shapes and settings.  Prints every mismatch beyond 1e-5 (onset: any)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
fx = importlib.import_module("feature-extractor_amd")
from oracle import fx_oracle as fo
from bench import usable_cores

THREADS = usable_cores()


from stress_signals import make_signal  # noqa: E402


def run(seconds=60.0, seed=0, max_cases=None, save_failures=True, verbose=True, windows=(256, 512, 1024, 1024, 2048, 2048, 4096), block_share=0.15):
    """Returns (cases, frames, mismatching cases, worst finite relative error)."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = frames = bad_cases = 0
    worst = 0.0
    inexact = np.zeros(12, np.int64)      # values that are within tolerance but not bit-identical, per slot
    ring_cases = [0]
    pair_cases = [0]
    hop_cases = [0]
    pcm_cases = [0]
    block_cases = [0]
    fed_cases = [0]
    block_frames = [0]
    last_note = time.time()
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        N = int(rng.choice(list(windows)))
        C = int(rng.integers(1, 24)) if rng.random() < 0.8 else int(rng.integers(24, 200))
        rt = rng.random()
        T = int(rng.integers(1, 40)) if rt < 0.85 else (int(rng.integers(40, 130)) if rt < 0.95 else int(rng.integers(128, 330)))
        if T >= 128:            # long calls are cut in time (several workgroups per channel, flux state handed on through memory)
            C = min(C, 16)
        order = int(rng.integers(0, 3))
        otype = int(rng.integers(0, 3))
        owin = int(rng.integers(1, 22))
        sens = float(rng.uniform(0, 2))
        gain = float(rng.choice([1.0, 1.0, 0.5, 3.0]))
        which = str(rng.choice(["both", "both", "both", "spectral", "harmonic"]))
        mask = {"both": 3, "spectral": 1, "harmonic": 2}[which]
        hops = make_signal(rng, C, T, N)
        feed = hops
        pick = rng.random()
        if pick < 0.2:
            # 16-bit PCM ingest (FX_SAMPLE_S16): the kernels widen v to v / 32768 in their load stage; the oracle analyses the decoded floats
            feed = np.clip(np.round(hops * 32768.0), -32768, 32767).astype(np.int16)
            hops = feed.astype(np.float32) / np.float32(32768.0)
            pcm_cases[0] += 1
        elif pick < 0.3:
            # packed 24-bit PCM (FX_SAMPLE_S24): three bytes per sample, v / 8388608
            v24 = np.clip(np.round(hops.astype(np.float64) * 8388608.0), -8388608, 8388607).astype(np.int32)
            feed = fx.pack_s24(v24)
            hops = v24.astype(np.float32) / np.float32(8388608.0)
            pcm_cases[0] += 1
        an = fx.BatchAnalyser(C, N, order=order, analysers=which)
        an.set_onset_detection_type(otype); an.set_onset_window_length(owin)
        an.set_onset_detection_sensitivity(sens); an.set_gain(gain)
        if which == "both" and N >= 2048 and rng.random() < 0.5:
            # one frame across a PAIR of wavefronts: fx_pair_kernel for the batch calls, fx_hop_pair_kernel for one-hop calls
            an.set_tuning(waves_per_frame=2)
            pair_cases[0] += 1
        split = int(rng.integers(0, T + 1))
        ring = which == "both" and N >= 1024 and T <= 24 and rng.random() < 0.25
        if ring:
            # one hop per call through the pinned ring: fx_hop_kernel (three wavefronts per channel, one launch per hop)
            st = fx.HopStream(an, 1, slots=3, dtype=feed.dtype)
            parts = []
            for t in range(T):
                if st.in_flight() == 3:
                    parts.append(st.collect())
                st.push(feed[:, t:t + 1])
            while st.in_flight():
                parts.append(st.collect())
            st.close()
            ring_cases[0] += 1
        elif rng.random() < block_share:
            # the collector's interface (fx_push_samples, round 5): the same stream as device blocks of random lengths -- an audio device's
            # 441 / 480 / 512, single samples, blocks longer than a window -- cut into hops on the device
            H = N // 2
            per = 3 if feed.dtype == np.uint8 else 1
            flat = np.ascontiguousarray(feed.reshape(C, -1))
            parts, at, total = [], 0, T * H
            # a third of these cases through the pinned ring's form (fx_stream_push_samples / fx_stream_collect_samples), up to three blocks in flight
            st = fx.HopStream(an, max(4097, 3 * N) // H + 2, slots=3, dtype=feed.dtype) if rng.random() < 0.33 else None
            # Round 6: a call that completes exactly one hop feeds the block to the one-frame kernels (FrameParams::block_mode).  Half of the
            # cases use an audio device's lengths only, so that nearly every call is such a call; which of the three kernels reads the block
            # (hop kernel / frames + tails in one launch / frame kernel then tail kernel) and where the block lives (host, device) is drawn too.
            live = rng.random() < 0.5
            on_device = st is None and per == 1 and rng.random() < 0.5
            kernel = int(rng.integers(0, 3))
            if kernel:
                an.set_tuning(one_hop_kernel=0)
                an.set_test_hooks(8 if kernel == 1 else 4)
            fed_cases[0] += live
            block_frames[0] += C * T
            if on_device:
                import torch
            while at < total:
                n = int(rng.choice([441, 480, 512, H, H - 1, H + 1, int(rng.integers(H // 2 + 1, H + H // 2))])) if live else \
                    int(rng.choice([1, 63, 441, 480, 512, 1000, 4097, int(rng.integers(1, 3 * N))]))
                if st is None and not live and rng.random() < 0.25:
                    n = int(rng.integers(3 * N, 70 * N))           # long blocks: at 1024 points one launch of the batch kernel's block-fed form, in units from ~48 hops
                n = min(n, total - at)
                piece = np.ascontiguousarray(flat[:, at * per:(at + n) * per])
                if on_device:
                    r_, s_ = an.push_samples(torch.from_numpy(piece).cuda())
                    parts.append((r_.cpu().numpy(), s_.cpu().numpy()))
                    at += n
                    continue
                if st is not None:
                    if st.in_flight() == 3:
                        parts.append(st.collect_samples())
                    st.push_samples(piece)
                else:
                    parts.append(an.push_samples(piece, sample_format="s24" if per == 3 else None))
                at += n
            if st is not None:
                while st.in_flight():
                    parts.append(st.collect_samples())
                st.close()
            assert an.pending_samples() == 0
            block_cases[0] += 1
        elif T <= 40 and rng.random() < 0.2:
            # hop by hop through fx_push_hops: fx_hop_kernel, or -- every other such case -- the batch kernels forced
            # (frame kernel + the one-frame form of fx_tail_fused_kernel: a lane per slot, the logarithms side by side)
            if hop_cases[0] % 2:
                an.set_tuning(one_hop_kernel=0)          # (keyword form: changes this field of the context's current knobs)
            hop_cases[0] += 1
            parts = [an.push_hops(feed[:, t:t + 1]) for t in range(T)]
        else:
            parts = [an.push_hops(feed[:, :split]), an.push_hops(feed[:, split:])]
        raw = np.concatenate([p[0] for p in parts], 1); sm = np.concatenate([p[1] for p in parts], 1)
        oraw, osm = fo.batch_hops(hops, N, order=order, threads=THREADS, onset_type=otype, onset_window=owin, onset_sensitivity=sens, gain=gain, analysers=mask)
        cases += 1; frames += C * T
        if verbose and time.time() - last_note > 60.0:          # (a run that prints nothing for minutes looks hung to the GPU box's watchdog)
            last_note = time.time()
            print("... %d cases, %d frames, %d mismatching so far" % (cases, frames, bad_cases), flush=True)
        for name, g, w in (("raw", raw, oraw), ("smoothed", sm, osm)):
            g64, w64 = g.astype(np.float64), w.astype(np.float64)
            same = (g64 == w64) | (np.isnan(g64) & np.isnan(w64))
            with np.errstate(invalid="ignore", divide="ignore"):
                err = np.where(same, 0.0, np.abs(g64 - w64) / np.abs(w64))
            err = np.where(np.isnan(err), np.inf, err)
            tol = np.full(12, 1e-5); tol[0] = 0.0
            bad = np.argwhere(err > tol)
            worst = max(worst, float(np.max(np.where(np.isfinite(err), err, 0))))
            if name == "raw":
                inexact += np.count_nonzero(~same, axis=(0, 1))
            if len(bad):
                bad_cases += 1
                if save_failures and bad_cases <= 6:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "stress_fail_%d.npz" % bad_cases), hops=hops, N=N, order=order,
                                        otype=otype, owin=owin, sens=sens, gain=gain, split=split, gpu_raw=raw, oracle_raw=oraw, which=name)
                c, t, f = bad[0]
                if verbose:
                    print("MISMATCH %s N=%d C=%d T=%d order=%d otype=%d owin=%d: %d values; first c=%d t=%d %s gpu=%r oracle=%r"
                          % (name, N, C, T, order, otype, owin, len(bad), c, t, fx.FEATURE_NAMES[f], g[c, t, f], w[c, t, f]), flush=True)
    if verbose:
        print("cases run one hop per call through the ring (fx_hop_kernel / fx_hop_pair_kernel): %d; cases on wavefront pairs (fx_pair_kernel): %d; "
              "cases hop by hop through fx_push_hops (half of them on the batch kernels + one-frame fused tail): %d; cases fed as 16- or 24-bit PCM: %d; "
              "cases fed as device blocks of random lengths through fx_push_samples: %d (%d frames; %d of the cases in an audio device's block lengths, "
              "where a call completes one hop and the one-frame kernels read the block themselves)"
              % (ring_cases[0], pair_cases[0], hop_cases[0], pcm_cases[0], block_cases[0], block_frames[0], fed_cases[0]), flush=True)
    if verbose and inexact.any():
        print("raw values not bit-identical (within tolerance), per slot:", dict((fx.FEATURE_NAMES[i], int(n)) for i, n in enumerate(inexact) if n), flush=True)
    return cases, frames, bad_cases, worst


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kw = dict(windows=tuple(int(v) for v in sys.argv[3].split(","))) if len(sys.argv) > 3 and sys.argv[3] != "-" else {}      # e.g. 4096  or  2048,4096
    if len(sys.argv) > 4:
        kw["block_share"] = float(sys.argv[4])            # share of the cases (after the ring's) fed as device blocks
    cases, frames, bad_cases, worst = run(seconds, seed, **kw)
    print("stress: %d cases, %d frames, %d mismatching cases, worst finite rel err %.3e" % (cases, frames, bad_cases, worst))


if __name__ == "__main__":
    main()
