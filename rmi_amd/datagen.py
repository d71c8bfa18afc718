"""Deterministic, integer-only synthetic key sets.

Every generator emits an already-sorted array so CPU and GPU see bit-identical input without
a sort, and writes/reads the reference's on-disk format: ``u64 LE count`` followed by
``count`` little-endian items, dtype selected by a substring of the file name
(reference: src/load.rs:132-157, src/main.rs:122-132).
"""
from __future__ import annotations

import numpy as np

_U64 = np.uint64
_MASK = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + _U64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _h(i: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(i.astype(np.uint64) + _U64(seed))


def uniform_u64(n: int, seed: int = 42, start: int = 0, count: int | None = None) -> np.ndarray:
    """key[i] = 1 + i*stride + (h(i) mod stride): strictly increasing, unique, in [1, 2^64-2].

    ``start``/``count`` select a contiguous index window of the *global* n-key array (used to
    shard generation across ranks)."""
    count = n - start if count is None else count
    stride = (_MASK - 1) // n
    i = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _U64(1) + i * _U64(stride) + (_h(i, seed) % _U64(stride))


def uniform_u32(n: int, seed: int = 46, start: int = 0, count: int | None = None) -> np.ndarray:
    count = n - start if count is None else count
    stride = ((1 << 32) - 3) // n
    if stride < 1:
        raise ValueError("n too large for unique u32 keys")
    i = np.arange(start, start + count, dtype=np.uint64)
    k = _U64(1) + i * _U64(stride) + (_h(i, seed) % _U64(stride))
    return k.astype(np.uint32)


def _apply_dups(keys: np.ndarray, seed: int) -> np.ndarray:
    """key[i] := key[i - (i mod r_i)], r_i in {1,1,1,2,8} chosen per group of 8 indices."""
    n = len(keys)
    i = np.arange(n, dtype=np.uint64)
    choice = (_h(i >> _U64(3), seed) % _U64(5)).astype(np.int64)
    r = np.array([1, 1, 1, 2, 8], dtype=np.uint64)[choice]
    src = (i - (i % r)).astype(np.int64)
    return keys[src]


def dups_u64(n: int, seed: int = 45) -> np.ndarray:
    return _apply_dups(uniform_u64(n, 42), seed)


def dups_u32(n: int, seed: int = 47) -> np.ndarray:
    return _apply_dups(uniform_u32(n, 46), seed)


def books_u64(n: int, seed: int = 43, segments: int = 4096) -> np.ndarray:
    """Heavy-tailed local density ("books_200M-shaped"): equal-count segments whose mean gap
    is 2^(20 + h(g) mod 14); unique, strictly increasing, total < 2^63."""
    segments = max(1, min(segments, n))
    i = np.arange(n, dtype=np.uint64)
    per = -(-n // segments)
    g = i // _U64(per)
    expo = _U64(20) + (_h(np.arange(segments, dtype=np.uint64), seed) % _U64(14))
    G = (_U64(1) << expo)
    cap = _U64(max(1, (1 << 62) // max(n, 1)))
    G = np.minimum(G, cap)
    Gi = G[g.astype(np.int64)]
    with np.errstate(over="ignore"):
        gap = _U64(1) + (_h(i, seed + 1) % (_U64(2) * Gi))
        return _U64(1) + np.cumsum(gap, dtype=np.uint64)


def clustered_u64(n: int, seed: int = 48, base: int = 1 << 62) -> np.ndarray:
    """Ill-conditioned set: tiny gaps on a huge offset (keys near 2^62 with gaps < 2^12), so
    f64(key) collapses many distinct keys; exercises the as-float rounding rules."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        gap = _U64(1) + (_h(i, seed) % _U64(4096))
        return _U64(base) + np.cumsum(gap, dtype=np.uint64)


def uniform_f64(n: int, seed: int = 49) -> np.ndarray:
    """f64 keys (file names containing "f64", src/main.rs:127-129): the 53 high bits of the uniform
    u64 set scaled by 2^-20 -- exact, monotone, non-integer values."""
    k = uniform_u64(n, seed) >> _U64(11)
    return k.astype(np.float64) * (2.0 ** -20)


# ---- adversarial sets for the one-pass mode's guard and for every floor(): exact linear structure puts every
# ---- prediction on an integer +- rounding (auto-increment ids, fixed-interval timestamps)
def progression_u64(n: int, stride: int = 10, start: int = 1) -> np.ndarray:
    """Arithmetic progression start + i stride (u64)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _U64(start) + i * _U64(stride)


def progression_u32(n: int, stride: int = 2, start: int = 1) -> np.ndarray:
    if start + (n - 1) * stride >= (1 << 32) - 1:
        raise ValueError("progression_u32 overflows")
    return (np.arange(n, dtype=np.uint64) * _U64(stride) + _U64(start)).astype(np.uint32)


def progression_f64(n: int, stride: float = 0.25, start: float = 1.0) -> np.ndarray:
    return start + np.arange(n, dtype=np.float64) * stride


def progression_outlier_u64(n: int, stride: int = 1 << 20) -> np.ndarray:
    """A progression with one far outlier at the end (a root that fits everything but one key)."""
    k = progression_u64(n, stride, 1)
    k[-1] = _U64((1 << 63) + 12345)
    return k


def around_2_53(n: int) -> np.ndarray:
    """Keys straddling 2^53, where key -> f64 starts to round (stride 3: odd and even keys)."""
    return progression_u64(n, 3, (1 << 53) - 3 * (n // 2))


def around_2_63(n: int) -> np.ndarray:
    """Stride-1025 keys straddling 2^63 (f64 spacing 1024 below, 2048 above)."""
    return progression_u64(n, 1025, (1 << 63) - 1025 * (n // 2))


GENERATORS = {
    "uniform_u64": uniform_u64,
    "books_u64": books_u64,
    "dups_u64": dups_u64,
    "uniform_u32": uniform_u32,
    "dups_u32": dups_u32,
    "clustered_u64": clustered_u64,
    "uniform_f64": uniform_f64,
}
ADVERSARIAL = {
    "prog1_u64": lambda n: progression_u64(n, 1),
    "prog10_u64": lambda n: progression_u64(n, 10),
    "prog2p20_u64": lambda n: progression_u64(n, 1 << 20),
    "prog2p44_u64": lambda n: progression_u64(n, 1 << 44) if n < (1 << 19) else progression_u64(n, (1 << 63) // n),
    "prog2_u32": lambda n: progression_u32(n, 2),
    "prog_f64": lambda n: progression_f64(n, 0.25),
    "prog_outlier_u64": progression_outlier_u64,
    "around_2_53": around_2_53,
    "around_2_63": around_2_63,
}


def write_keys(path: str, keys: np.ndarray) -> None:
    """Reference data-file format (README.md:26-31; src/load.rs:140)."""
    with open(path, "wb") as f:
        f.write(np.array([len(keys)], dtype="<u8").tobytes())
        f.write(np.ascontiguousarray(keys).astype(keys.dtype.newbyteorder("<"), copy=False).tobytes())


def dtype_from_path(path: str):
    """src/main.rs:122-132: dtype by substring of the path, in this order."""
    if "uint64" in path:
        return np.dtype("<u8")
    if "uint32" in path:
        return np.dtype("<u4")
    if "f64" in path:
        return np.dtype("<f8")
    raise ValueError("Data file must contain uint64, uint32, or f64.")


def read_keys(path: str) -> np.ndarray:
    dt = dtype_from_path(path)
    with open(path, "rb") as f:
        n = int(np.frombuffer(f.read(8), dtype="<u8")[0])
    return np.memmap(path, dtype=dt, mode="r", offset=8, shape=(n,))


def books_u64_torch(n: int, device="cuda", seed: int = 43, segments: int = 4096):
    """``books_u64`` produced in HBM with torch (int64 tensors carrying the uint64 bit patterns: wrapping adds and
    multiplies are the same bits, logical shifts are masked arithmetic ones): 200 M keys in tens of milliseconds
    instead of a minute of numpy on the host.  Valid where every segment's mean gap is a power of two below the cap
    (n <= 2^29), so that ``h mod 2G`` is a bit mask; bit-identical to ``books_u64`` (tests/test_gpu_parity.py)."""
    import torch
    if n > (1 << 29):
        raise ValueError("books_u64_torch: n <= 2^29")
    segments = max(1, min(segments, n))

    def i64(v):                                  # python int (uint64 bit pattern) -> the same bits as a signed value
        v &= _MASK
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, s):                               # logical shift right of int64 bit patterns
        return (z >> s) & ((1 << (64 - s)) - 1)

    def smix(x):
        z = x + i64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
        return z ^ lsr(z, 31)

    per = -(-n // segments)
    sg = torch.arange(segments, dtype=torch.int64, device=device)
    # (h mod 14 of the unsigned 64-bit pattern: (hi * 2^32 + lo) mod 14 from two non-negative halves)
    h = smix(sg + seed)
    hi, lo = lsr(h, 32), h & 0xFFFFFFFF
    expo = 20 + (hi * ((1 << 32) % 14) + lo) % 14
    cap = max(1, (1 << 62) // max(n, 1))
    if (1 << 33) > cap:
        raise ValueError("books_u64_torch: a segment's mean gap would hit the cap")
    mask2g = (torch.ones_like(expo) << (expo + 1)) - 1                     # 2 G - 1
    i = torch.arange(n, dtype=torch.int64, device=device)
    gap = 1 + (smix(i + (seed + 1)) & mask2g[torch.div(i, per, rounding_mode="floor")])
    return 1 + torch.cumsum(gap, dim=0)                                    # (total < 2^63: no wrap)
