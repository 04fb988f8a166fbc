"""NumPy Philox4x32-10 (test infrastructure): independent of csrc/philox.h, used to check the
device RNG draws (episode starts, random pre-fill actions, exploration noise, minibatch indices)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
STREAM_RESET, STREAM_RANDACT, STREAM_NOISE, STREAM_SAMPLE, STREAM_INIT = (
    0x52455345, 0x52414354, 0x4E4F4953, 0x53414D50, 0x494E4954)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & 0xFFFFFFFF for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = np.uint64(k0 & 0xFFFFFFFF)
    k1 = np.uint64(k1 & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def u01_24(x):
    return (np.asarray(x, np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def reset_draws(seed, episode, n, nrow, maxsteps, soc_max):
    i = np.arange(n, dtype=np.uint64)
    x, y, _, _ = philox4x32_10(i & 0xFFFFFFFF, i >> np.uint64(32), episode, STREAM_RESET, seed & 0xFFFFFFFF, seed >> 32)
    idx0 = 1 + (x % np.uint32(nrow - maxsteps)).astype(np.int32)
    soc0 = u01_24(y) * np.float32(soc_max)
    return idx0, soc0.astype(np.float32)


def random_actions(seed, step, n):
    i = np.arange(n, dtype=np.uint64)
    x, y, _, _ = philox4x32_10(i & 0xFFFFFFFF, i >> np.uint64(32), step, STREAM_RANDACT, seed & 0xFFFFFFFF, seed >> 32)
    a0 = (x.astype(np.float64) * (1.0 / 4294967296.0) * 2.0 - 1.0).astype(np.float32)
    a1 = (y.astype(np.float64) * (1.0 / 4294967296.0) * 2.0 - 1.0).astype(np.float32)
    return np.stack([a0, a1], 1)
