"""numpy restatement of the counter-based dropout generator (csrc/cr_common.hpp) so parity tests can run
the oracle with exactly the masks the kernels use."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    h = np.asarray(h, np.uint64) & M32
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def site_key(seed, step, site):
    inner = (np.uint64(step) * np.uint64(0x9E3779B9) + np.uint64(site) * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & M32
    return fmix32(np.uint64(seed) ^ fmix32(inner))


def thresh_scale(rate):
    r32 = np.float32(rate)
    t = float(r32) * 4294967296.0
    thresh = 4294967295 if t >= 4294967295.0 else int(t)
    return np.uint64(thresh), np.float32(1.0) / (np.float32(1.0) - r32)


def keep_mask(seed, step, site, rate, idx):
    """idx: integer array of element indices -> bool keep mask."""
    key = site_key(seed, step, site)
    thresh, _ = thresh_scale(rate)
    x = (np.asarray(idx, np.uint64) * np.uint64(0x9E3779B1) + key) & M32          # Weyl counter (cr_common.hpp CR_PHI)
    h = ((x ^ (x >> np.uint64(16))) * np.uint64(0xD168AAAD)) & M32                # cr_mix
    return h >= thresh


def rows_mask(seed, step, site, rate, M, N, row_offset=0):
    idx = (np.arange(M, dtype=np.uint64)[:, None] + np.uint64(row_offset)) * np.uint64(N) + np.arange(N, dtype=np.uint64)[None]
    return keep_mask(seed, step, site, rate, idx & M32)


def attn_mask(seed, step, site, rate, H, B, T, batch_global=None, row_offset=0):
    """[H*B, T, T] keep mask; row j*B + n (modules.py:208-213 head-major layout)."""
    Bg = B if batch_global is None else batch_global
    n0 = row_offset // T
    j = np.arange(H, dtype=np.uint64)[:, None, None, None]
    n = np.arange(B, dtype=np.uint64)[None, :, None, None] + np.uint64(n0)
    q = np.arange(T, dtype=np.uint64)[None, None, :, None]
    k = np.arange(T, dtype=np.uint64)[None, None, None, :]
    idx = ((j * np.uint64(Bg) + n) * np.uint64(T) + q) * np.uint64(T) + k
    return keep_mask(seed, step, site, rate, idx & M32).reshape(H * B, T, T)
