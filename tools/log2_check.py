"""Host emulation of sc_log2_int (rmi_scan.hip.h): the same operation sequence in numpy doubles against numpy's log2."""
import numpy as np
Lg = [6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01, 1.818357216161805012e-01,
      1.531383769920937332e-01, 1.479819860511658591e-01]


def fast_log2(v):
    v = np.asarray(v, dtype=np.float64)
    m, e = np.frexp(v)
    m = m * 2
    e = e - 1
    big = m > 1.4142135623730951
    m = np.where(big, m * 0.5, m)
    e = np.where(big, e + 1, e)
    f = m - 1.0
    d = 2.0 + f
    r = (1.0 / d).astype(np.float32).astype(np.float64)          # v_rcp_f64 is better than this
    r = r * (2.0 - d * r)
    r = r * (2.0 - d * r)
    s = f * r
    z = s * s
    w = z * z
    t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]))
    t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])))
    hfsq = 0.5 * f * f
    return e + (f - (hfsq - s * (hfsq + (t1 + t2)))) * 1.4426950408889634


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    vs = np.concatenate([np.arange(2, 200002, 2, dtype=np.float64), 2 * rng.integers(1, 2 ** 32, size=200000).astype(np.float64) + 2, [2.0 ** 33, 2.0 ** 33 - 2]])
    ref = np.log2(vs)
    rel = np.abs(fast_log2(vs) - ref) / ref
    print("max relative error", rel.max())
