"""The launches bench.py times, held to the oracle and to the uncut launch; the hand-over between work units failing
loudly; one-frame calls through the one-launch hop kernel; contexts used from two threads.  All through the C ABI."""
import threading

import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, want, what):
    from oracle import fx_oracle as fo
    return signals.assert_features_close(got, want, RTOL, fo.FEATURE_NAMES, what)


@pytest.mark.parametrize("C,T", [(1024, 512), (8192, 128)], ids=["bench_1gpu_1024x512", "bench_per_rank_8192x128"])
def test_bench_launch_matches_oracle_and_the_uncut_launch(gpu_fx, oracle, C, T):
    """Exactly the shapes bench.py times (configs[1]: 1024 channels x 512 frames x 1024-pt, nine work units of decreasing
    length per channel; configs[3] per rank: 8192 x 128, two units), through fx_process_frames on device-resident input:
    24 random channels against the oracle, every value against the launch that is not cut in time (one workgroup per
    channel), and a second call that continues from the first one's state (flux state, window tail, histories:
    ref SpectralCharacteristics.h:76-79,121-123,138,203)."""
    import torch
    N = 1024
    rng = np.random.default_rng(C + T)
    sel = np.sort(rng.choice(C, 24, replace=False))
    host = gpu_fx.synth.frames(C, T, N)
    dev = torch.from_numpy(host).cuda()
    # second call: other contents without a second trip through the generator -- every channel gets half of the frames
    # of the channel 37 below it (x 0.5 is exact), with a stretch of digital silence across unit boundaries in six of the
    # checked channels: the skip rule (:121-123) at a hand-over
    second_host = 0.5 * host[(sel - 37) % C]
    second_host[:6, 40:90] = 0.0
    dev2 = dev.roll(37, dims=0) * 0.5
    dev2[torch.from_numpy(sel[:6]).cuda(), 40:90] = 0.0
    calls = [(dev, host[sel].copy()), (dev2, second_host)]
    del host
    an = gpu_fx.BatchAnalyser(C, N)
    plan = gpu_fx.capi.plan_units(N, 0, 8, T, an.get_tuning())
    assert len(plan) >= 2, plan                         # the launch under test really is cut in time
    an0 = gpu_fx.BatchAnalyser(C, N)
    an0.set_tuning(frames_per_unit=0)
    want_frames = np.concatenate([c[1] for c in calls], axis=1)
    oraw, osm = oracle.process_frames(want_frames, N)
    for k, (dev, _) in enumerate(calls):
        raw, sm = an.process_frames(dev)
        raw0, sm0 = an0.process_frames(dev)
        an.sync(); an0.sync()
        assert torch.equal(torch.nan_to_num(raw), torch.nan_to_num(raw0)) and torch.equal(raw.isnan(), raw0.isnan()), "call %d raw: cut != uncut" % k
        assert torch.equal(torch.nan_to_num(sm), torch.nan_to_num(sm0)) and torch.equal(sm.isnan(), sm0.isnan()), "call %d smoothed: cut != uncut" % k
        close(raw[sel].cpu().numpy(), oraw[:, k * T:(k + 1) * T], "bench launch call %d raw" % k)
        close(sm[sel].cpu().numpy(), osm[:, k * T:(k + 1) * T], "bench launch call %d smoothed" % k)
    assert np.array_equal(an.get_features(), an0.get_features(), equal_nan=True)


def test_failed_handover_between_work_units_is_reported(gpu_fx):
    """A work unit that gives up waiting for its predecessor's flux state must not carry on silently (it would hand a
    stale previousBinMagnitudes down the channel, ref SpectralCharacteristics.h:76-79,203): the kernel raises an error
    word and every synchronising entry point returns FX_ERR_HIP until fx_reset_state.  Forced here by units that do
    not publish (fx_set_tuning_internal bit 0) and a poll bound of 8."""
    C, T, N = 64, 256, 1024
    hops = signals.bursts(C, T, N, seed=77)
    good = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    an = gpu_fx.BatchAnalyser(C, N)
    an.set_tuning(handover_spin_limit=8); an.set_test_hooks(1)
    with pytest.raises(gpu_fx.FxError) as e:
        an.push_hops(hops)                              # host buffers: the call synchronises and must report
    assert e.value.code == 3 and "timed out" in str(e.value)
    with pytest.raises(gpu_fx.FxError):
        an.sync()                                       # sticky
    with pytest.raises(gpu_fx.FxError):
        an.push_hops(hops[:, :4])
    an.reset_state()                                    # a fresh analyser again
    an.set_tuning(handover_spin_limit=0); an.set_test_hooks(0)
    got = an.push_hops(hops)
    an.sync()
    for k in (0, 1):
        assert np.array_equal(got[k], good[k], equal_nan=True)


@pytest.mark.parametrize("N,C", [(256, 37), (512, 9), (1024, 21), (1024, 1), (2048, 7), (4096, 10)])
def test_one_frame_calls_finish_their_tail_in_the_frame_kernel(gpu_fx, N, C):
    """One frame per channel through the batch kernels is ONE launch while the chip holds the call's workgroups at once
    (fx_frame_tail_kernel: the workgroup's first wavefronts finish its channels' hops when the frames are done), else the frame
    kernel and the fused tail kernel (windows of 1024 points and more: below that the one-launch form is not built).  fx_set_tuning_internal bits 3 / 2 force the one and the other: the same bits, hop after hop, with an onset window that reaches far into the history ring, channel
    counts that leave the last workgroup partly filled, and device buffers."""
    import torch
    T = 60
    hops = np.concatenate([signals.bursts(C, T // 2, N, seed=N + C), signals.tone_vibrato_noise(C, T - T // 2, N, seed=C)], axis=1)
    one, two = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    one.set_test_hooks(8)
    two.set_test_hooks(4)
    for an in (one, two):
        an.set_tuning(one_hop_kernel=0)
        an.set_onset_window_length(21); an.set_onset_detection_type(2); an.set_onset_detection_sensitivity(0.4)
    dev = torch.from_numpy(hops).cuda()
    for t in range(T):
        a = one.push_hops(dev[:, t:t + 1].contiguous())
        b = two.push_hops(hops[:, t:t + 1])
        for k in (0, 1):
            assert np.array_equal(a[k].cpu().numpy(), b[k], equal_nan=True), (t, k)
    assert np.array_equal(one.get_features(), two.get_features(), equal_nan=True)
    whole = gpu_fx.BatchAnalyser(C, N)
    whole.set_onset_window_length(21); whole.set_onset_detection_type(2); whole.set_onset_detection_sensitivity(0.4)
    want = whole.push_hops(hops)
    assert np.array_equal(one.get_features(), want[1][:, -1], equal_nan=True)


@pytest.mark.parametrize("ch_per_wg", [3, 5, 6, 7])
def test_one_frame_calls_with_a_channel_count_per_workgroup_that_is_no_multiple_of_four(gpu_fx, ch_per_wg):
    """fx_frame_tail_kernel's tail wavefronts take four channels each: with 3, 5, 6 or 7 channels per workgroup the last group of a
    workgroup's last tail wavefront belongs to the NEXT workgroup and must be left alone (round-4 advisor finding: it used to read that
    channel's record, possibly before it was written, and both workgroups wrote its history)."""
    C, N, T = 23, 1024, 40
    hops = np.concatenate([signals.bursts(C, T // 2, N, seed=ch_per_wg), signals.tone_vibrato_noise(C, T - T // 2, N, seed=3)], axis=1)
    one, two = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    one.set_test_hooks(8)                      # always finish the hop's tail in the frame kernel
    one.set_tuning(one_hop_kernel=0, channels_per_workgroup=ch_per_wg)
    two.set_test_hooks(4)                      # never
    two.set_tuning(one_hop_kernel=0)
    for t in range(T):
        a, b = one.push_hops(hops[:, t:t + 1]), two.push_hops(hops[:, t:t + 1])
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True), t
    assert np.array_equal(one.get_features(), two.get_features(), equal_nan=True)


def test_short_calls_cut_into_work_units_leave_the_ticket_counter_at_zero(gpu_fx):
    """A call of <= 8 frames per channel finishes with fx_tail_fused_kernel; cut into work units by an explicit plan it must still
    leave the units' ticket counter and hand-over counts zeroed for the next cut call (round-4 advisor finding: only fx_history_kernel
    cleared them, so the second such call started its tickets past the plan and analysed nothing)."""
    C, N, T = 6, 1024, 8
    hops = signals.tone_vibrato_noise(C, 4 * T, N, seed=11)
    cut, whole = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    cut.set_tuning(waves_per_channel=2, unit_plan=[4, 4])
    whole.set_tuning(frames_per_unit=0)
    for k in range(4):
        a, b = cut.push_hops(hops[:, k * T:(k + 1) * T]), whole.push_hops(hops[:, k * T:(k + 1) * T])
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True), k
    # ... and a long cut call right after a short one
    more = signals.tone_vibrato_noise(C, 200, N, seed=12)
    cut.set_tuning(waves_per_channel=0, unit_plan=[])
    whole.set_tuning(waves_per_channel=0)
    a, b = cut.push_hops(more), whole.push_hops(more)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True)


def test_4096_twiddle_fallback_gives_the_same_bits(gpu_fx):
    """The 4096-point frame kernel keeps 24 KB of its 32 KB of twiddles in LDS and forms two rows of the last pass as quarter
    turns of two others, which the float table allows on this host (fx_create checks it entry by entry).  A host where it
    does not reads those rows from the whole table in global memory: forced here (fx_set_tuning_internal bit 1), batch calls
    and one-frame calls (the kernel that leaves the flux state in global memory) -- same values, so the same bits."""
    import torch
    C, T, N = 20, 19, 4096
    hops = np.concatenate([signals.bursts(C, 10, N, seed=5), signals.low_tones(C, T - 10, N)], axis=1)
    a, b = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    b.set_test_hooks(2)
    for an in (a, b):
        an.set_tuning(one_hop_kernel=0)
    ra, rb = a.push_hops(hops[:, :12]), b.push_hops(hops[:, :12])
    for k in (0, 1):
        assert np.array_equal(ra[k], rb[k], equal_nan=True)
    for t in range(12, T):                               # one frame per call through the batch kernels
        ra, rb = a.push_hops(hops[:, t:t + 1]), b.push_hops(hops[:, t:t + 1])
        for k in (0, 1):
            assert np.array_equal(ra[k], rb[k], equal_nan=True)
    assert ra[0][:, 0, 2].max() > 0                      # (F0 slot: the analysers did run)


@pytest.mark.parametrize("N,C", [(1024, 300), (2048, 130), (4096, 40)])
def test_one_frame_calls_run_the_hop_kernel_and_equal_the_batch_kernels(gpu_fx, oracle, N, C):
    """One frame per channel per call -- the reference's own cadence (ref AudioDataCollector.h:66-94, RealTimeAnalyser.h:
    201-234) -- runs as ONE launch of fx_hop_kernel from fx_push_hops / fx_process_frames too (host or device buffers, any
    channel count): bit for bit what the batch kernels give, and the oracle's values."""
    import torch
    T = 14
    hops = np.concatenate([signals.bursts(C, T // 2, N, seed=N), signals.low_tones(C, T - T // 2, N)], axis=1)
    hop_k, batch_k = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    batch_k.set_tuning(one_hop_kernel=0)
    dev = torch.from_numpy(hops).cuda()
    got, want = [], []
    for t in range(T):
        if t % 2:
            r, s = hop_k.push_hops(dev[:, t:t + 1].contiguous())
            got.append((r.cpu().numpy(), s.cpu().numpy()))
        else:
            got.append(hop_k.push_hops(hops[:, t:t + 1]))
        want.append(batch_k.push_hops(hops[:, t:t + 1]))
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), np.concatenate([w[k] for w in want], 1), equal_nan=True), k
    assert np.array_equal(hop_k.get_features(), batch_k.get_features(), equal_nan=True)
    sel = np.arange(0, C, max(1, C // 12))
    oraw, osm = oracle.push_hops(hops[sel], N)
    close(np.concatenate([g[0] for g in got], 1)[sel], oraw, "one-frame calls raw")
    close(np.concatenate([g[1] for g in got], 1)[sel], osm, "one-frame calls smoothed")
    # pre-assembled windows, one per call
    frames = gpu_fx.synth.frames(C, 5, N, first_channel=3)
    a, b = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    b.set_tuning(one_hop_kernel=0)
    for t in range(5):
        ga, gb = a.process_frames(frames[:, t:t + 1]), b.process_frames(frames[:, t:t + 1])
        assert np.array_equal(ga[0], gb[0], equal_nan=True) and np.array_equal(ga[1], gb[1], equal_nan=True), t


@pytest.mark.parametrize("N,C,onset", [(1024, 1100, (8, 2)), (4096, 1100, (5, 1))])
def test_large_one_frame_calls_take_the_batch_kernels_and_equal_the_hop_kernel(gpu_fx, oracle, N, C, onset):
    """Above 2^20 samples per call (2^22 at 4096 points) a one-frame call runs as the frame kernel + fx_tail_fused_kernel, whose one-frame form gives every
    smoothed slot a lane, evaluates the onset detector's candidates side by side and takes the scalar tail's logarithms in five lanes:
    bit for bit what fx_hop_kernel (forced) gives, and the oracle's values -- onsets included, over enough hops to fill every history."""
    assert C * N > ((1 << 22) if N == 4096 else (1 << 20))
    T = 26
    hops = np.concatenate([signals.bursts(C, T // 2, N, seed=7 * N), signals.tone_vibrato_noise(C, T - T // 2, N, seed=N)], axis=1)
    auto, hop_k = gpu_fx.BatchAnalyser(C, N), gpu_fx.BatchAnalyser(C, N)
    hop_k.set_tuning(one_hop_kernel=1)
    for an in (auto, hop_k):
        an.set_onset_window_length(onset[0]); an.set_onset_detection_type(onset[1]); an.set_onset_detection_sensitivity(0.3)
    got = [auto.push_hops(hops[:, t:t + 1]) for t in range(T)]
    want = [hop_k.push_hops(hops[:, t:t + 1]) for t in range(T)]
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k] for g in got], 1), np.concatenate([w[k] for w in want], 1), equal_nan=True), k
    assert np.array_equal(auto.get_features(), hop_k.get_features(), equal_nan=True)
    sel = np.arange(0, C, C // 10 if N < 4096 else C // 5 + 1)
    oraw, osm = oracle.push_hops(hops[sel], N, onset_window=onset[0], onset_type=onset[1], onset_sensitivity=0.3)
    close(np.concatenate([g[0] for g in got], 1)[sel], oraw, "large one-frame calls raw")
    close(np.concatenate([g[1] for g in got], 1)[sel], osm, "large one-frame calls smoothed")
    assert np.array_equal(np.concatenate([g[0] for g in got], 1)[sel][..., 0], oraw[..., 0])          # onsets, exactly


def test_calls_on_the_library_stream_need_no_cross_stream_waits(gpu_fx):
    """`with torch.cuda.stream(analyser.torch_stream())`: producer, analysis and consumer are ordered by the stream itself; device-buffer
    calls then skip their waits against torch's current stream and give what the ordinary path gives."""
    import torch
    N, C, T = 1024, 9, 12
    hops = signals.tone_vibrato_noise(C, T, N, seed=5)
    want = gpu_fx.BatchAnalyser(C, N).push_hops(hops)
    an = gpu_fx.BatchAnalyser(C, N)
    got = []
    with torch.cuda.stream(an.torch_stream()):
        assert torch.cuda.current_stream().cuda_stream == an.torch_stream().cuda_stream
        dev = torch.from_numpy(hops).cuda(non_blocking=False)
        for t in range(T):
            x = (dev[:, t:t + 1] * 1.0).contiguous()                 # produced on the library's stream, right before the call
            r, s = an.push_hops(x)
            got.append((r.clone(), s.clone()))                       # consumed on it, right after
    an.sync()
    for k in (0, 1):
        assert np.array_equal(np.concatenate([g[k].cpu().numpy() for g in got], 1), want[k], equal_nan=True), k


def test_one_frame_calls_record_timing_events_only_when_asked(gpu_fx):
    """fx_last_kernel_ms() reads three HIP events around the launches; each is a barrier packet, which halves the rate of back-to-back
    one-frame calls -- so those record none by default (fx_tuning::call_timing), and fx_last_kernel_ms says so instead of returning a stale time."""
    N, C = 1024, 5
    hops = signals.tone_vibrato_noise(C, 6, N, seed=2)
    an = gpu_fx.BatchAnalyser(C, N)
    an.push_hops(hops[:, :3])
    assert an.last_kernel_ms()[0] > 0.0                              # a call of several frames is timed
    an.push_hops(hops[:, 3:4])
    with pytest.raises(gpu_fx.FxError):
        an.last_kernel_ms()
    an.set_tuning(call_timing=1)
    an.push_hops(hops[:, 4:5])
    assert an.last_kernel_ms()[0] > 0.0
    an.set_tuning(call_timing=0)
    an.push_hops(hops[:, 5:6])
    with pytest.raises(gpu_fx.FxError):
        an.last_kernel_ms()


def test_contexts_created_and_streamed_from_two_threads(gpu_fx):
    """Two contexts on device 0, each created and driven by its own thread, streaming one hop per call (fx_hop_kernel
    through the pinned ring): kernel preparation is per context / device, with no process-wide "already prepared" state
    to race on.  Both must give what a single-threaded run gives."""
    N, C, T = 2048, 6, 30
    hops = [signals.tone_vibrato_noise(C, T, N, seed=s) for s in (41, 42)]
    want = [gpu_fx.BatchAnalyser(C, N).push_hops(h) for h in hops]
    out, errs = [None, None], []

    def worker(k):
        try:
            an = gpu_fx.BatchAnalyser(C, N)
            st = gpu_fx.HopStream(an, 1, slots=2)
            got = []
            for t in range(T):
                if st.in_flight() == 2:
                    got.append(st.collect())
                st.push(hops[k][:, t:t + 1])
            while st.in_flight():
                got.append(st.collect())
            st.close()
            an.close()
            out[k] = (np.concatenate([g[0] for g in got], 1), np.concatenate([g[1] for g in got], 1))
        except Exception as e:          # noqa: BLE001 -- reported below, in the main thread
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs, errs
    for k in (0, 1):
        assert np.array_equal(out[k][0], want[k][0], equal_nan=True) and np.array_equal(out[k][1], want[k][1], equal_nan=True), k
