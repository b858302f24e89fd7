"""The cases of tests/golden/blocks/startup.npz (the reference stepped as the application steps it), shared by the generator and the tests."""
import numpy as np

import signals

CASES = [
    # name, N, C, total samples, device block, order
    ("app_2048_512", 2048, 2, 9000, 512, 0),      # the app's defaults: 4 hops of zeros, then every stretch of the ring twice
    ("app_1024_480", 1024, 2, 9000, 480, 1),      # 1 hop of zeros, the first 512 samples never analysed
    ("app_2048_441", 2048, 2, 12000, 441, 0),     # 4 hops of zeros, then the stream in order
    ("app_1024_512", 1024, 2, 9000, 512, 2),      # 8 hops of zeros (a whole lap), then the stream in order
]


def stream_of(N, C, total, seed):
    hops = -(-total // (N // 2))
    x = np.concatenate([signals.bursts(C, hops // 2, N, seed=seed), signals.tone_vibrato_noise(C, hops - hops // 2, N, seed=seed + 1)], axis=1).reshape(C, -1)
    return np.ascontiguousarray(x[:, :total], np.float32)
