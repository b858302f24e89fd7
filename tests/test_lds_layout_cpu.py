"""The paddings of the per-wave LDS images (csrc/fx_fft.hip.h: rimg, bimg, cpad, qpad) against the gfx950 bank model
(tools/lds_conflicts.py, MI355X guide's LDS table): the access patterns they were chosen for must be conflict-free.
CPU only: this pins the layout arithmetic, the GPU parity tests pin the kernels that use it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from lds_conflicts import cycles  # noqa: E402


def geo(N):
    P, U = N // 64, N // 128
    RQ = max(P, 16)
    BQ = U if U >= 8 else 0
    return P, U, RQ, BQ


def rimg(N, n):
    return n + 4 * (n // geo(N)[2])


def bimg(N, b):
    BQ = geo(N)[3]
    return b + 4 * (b // BQ) if BQ else b


def cpad(p):
    return p + (p >> 4)


def qpad(N, q):
    return q + 8 * (q >> 6) if N == 2048 else q


def free(kind, addrs):
    c, base = cycles(kind, addrs)
    return c == base


@pytest.mark.parametrize("N", [1024, 2048, 4096])
def test_lowpass_runs_are_conflict_free(N):
    """lane l reads / writes its own run of N/64 samples 16 bytes at a time (lowpass_window)"""
    P = N // 64
    for i in range(0, P, 4):
        assert free("read_b128", [4 * rimg(N, P * l + i) for l in range(64)])
        assert free("write_b128", [4 * rimg(N, P * l + i) for l in range(64)])


@pytest.mark.parametrize("N", [2048, 4096])
def test_frame_store_is_conflict_free(N):
    """load_window: 16 bytes per lane, consecutive lanes, 256 samples per instruction"""
    for q in range(N // 256):
        assert free("write_b128", [4 * rimg(N, 256 * q + 4 * l) for l in range(64)])


@pytest.mark.parametrize("N", [1024, 2048, 4096])
def test_bins_runs_are_conflict_free(N):
    """lane l reads its U bins 16 bytes at a time (spectral / harmonic sums, flux state)"""
    U = N // 128
    for j in range(0, U, 4):
        assert free("read_b128", [4 * bimg(N, U * l + j) for l in range(64)])


@pytest.mark.parametrize("N", [1024, 2048, 4096])
def test_last_pass_bin_stores_cost_nothing_extra(N):
    """consecutive lanes store consecutive bins (one dword each): at most 2-way, which ds_write_b32 absorbs"""
    for g in range(2):
        c, base = cycles("write_b32", [4 * bimg(N, l + 64 * g) for l in range(64)])
        assert c <= 2 * base


@pytest.mark.parametrize("N", [2048, 4096])
def test_split_exchanges_are_conflict_free(N):
    RA = 16 if N == 4096 else 8
    L1 = 16 if N == 4096 else 8
    L2 = N // 16
    # first exchange, first-pass stores: item lane + 64*g' at (lane + 64*g')*RA + i
    for i in (0, 1, RA - 1):
        assert free("write_b64", [8 * cpad(l * RA + i) for l in range(64)])
    # first exchange, second-pass loads: item it = lane, elements at stride L1
    def item_off(L0, i):
        return L0 * i + ((L0 // 16) * i if L0 >= 16 else ((L0 * i) >> 4))
    for i in range(16):
        assert free("read_b64", [8 * (cpad((l // L1) * (16 * L1) + l % L1) + item_off(L1, i)) for l in range(64)])
    # second exchange: stores of element i (< 8) of item lane + 64*j, loads of row i by 64 consecutive lanes
    for j in range((N // 16) // 64):
        for i in range(8):
            assert free("write_b64", [8 * (qpad(N, (L2 // 2) * ((l + 64 * j) // L1)) + (l + 64 * j) % L1 + L1 * i) for l in range(64)])
    for gl in range(((N // 16) // 64) // 2):
        for i in range(16):
            assert free("read_b64", [8 * (l + 64 * gl + qpad(N, (L2 // 2) * i)) for l in range(64)])
