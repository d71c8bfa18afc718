"""Calibrate the guard bound of the sufficient-statistics mode against the oracle (CPU only)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import binding as orc
from rmi_amd import datagen

lib = C.CDLL("/tmp/libsigma_model.so")
P = C.c_void_p

def run(gen, n, L, tile=4096):
    keys = datagen.GENERATORS[gen](n)
    t0 = time.time()
    o = orc.train_two_layer("linear", "linear", keys, L, threads=2)
    t1 = time.time()
    ls = np.ascontiguousarray(o.leaf_start)
    # split idx
    mid = L // 2
    split_idx = int(ls[mid]) if ls[mid] < n else n   # first index with target >= L/2 (leaf_start is filled)
    psig = np.zeros((L, 2)); info = np.zeros((L, 5)); reg = np.zeros(L, dtype=np.uint8)
    lib.sigma_fit(P(keys.ctypes.data), C.c_uint64(n), P(ls.ctypes.data), C.c_uint64(L), C.c_uint64(tile), C.c_uint64(split_idx),
                  P(psig.ctypes.data), P(info.ctypes.data), P(reg.ctypes.data))
    pref = np.ascontiguousarray(o.leaf_params)
    disc = np.zeros(L); closest = np.zeros(L); er = np.zeros(L, dtype=np.uint64); es = np.zeros(L, dtype=np.uint64)
    lib.sigma_eval(P(keys.ctypes.data), C.c_uint64(n), P(ls.ctypes.data), C.c_uint64(L), P(pref.ctypes.data), P(psig.ctypes.data),
                   P(reg.ctypes.data), P(disc.ctypes.data), P(closest.ctypes.data), P(er.ctypes.data), P(es.ctypes.data))
    r = reg.astype(bool)
    print(f"{gen} n={n} L={L}: oracle {t1-t0:.1f}s, regular leaves {r.sum()} / nonempty {(ls[1:]>ls[:-1]).sum()}")
    cnt, X, W, sg = info[r, 0], info[r, 1], info[r, 2], info[r, 3]
    beta = np.abs(psig[r, 1]); Y = ls[1:][r].astype(np.float64)
    u = 2.0 ** -53
    t1_ = cnt * beta * X * u
    t2_ = cnt * beta * W * X / sg * u
    t3_ = (beta * X + Y) * u + 4 * info[r, 4] * beta * W * u
    d = disc[r]
    rel = np.abs(psig[r] - pref[r]) / np.maximum(np.abs(pref[r]), 1e-300)
    print("  max rel coeff diff alpha %.3e beta %.3e" % (rel[:, 0].max(), rel[:, 1].max()))
    print("  disc: max %.3e  median %.3e  p99.9 %.3e" % (d.max(), np.median(d), np.quantile(d, 0.999)))
    for name, bound in (("lin", t1_ + t2_ + t3_), ("sqrt", (t1_ + t2_) / np.sqrt(cnt) + t3_)):
        ratio = d / bound
        print(f"  bound[{name}]: median {np.median(bound):.3e}; disc/bound max {ratio.max():.3f} p99.9 {np.quantile(ratio,0.999):.3f}")
        for K in (1, 2, 4, 8, 16):
            flagged = closest[r] < K * bound
            mism = (er[r] != es[r]) & ~flagged
            print(f"    K={K}: flagged {flagged.mean()*100:.3f}%  unflagged err mismatches {mism.sum()}  (total mismatches {(er[r]!=es[r]).sum()})")
    return

if __name__ == "__main__":
    gen = sys.argv[1]; n = int(sys.argv[2]); L = int(sys.argv[3])
    run(gen, n, L)

def top(gen, n, L, tile=4096, k=8):
    keys = datagen.GENERATORS[gen](n)
    o = orc.train_two_layer("linear", "linear", keys, L, threads=2)
    ls = np.ascontiguousarray(o.leaf_start)
    mid = L // 2
    split_idx = int(ls[mid]) if ls[mid] < n else n
    psig = np.zeros((L, 2)); info = np.zeros((L, 5)); reg = np.zeros(L, dtype=np.uint8)
    lib.sigma_fit(P(keys.ctypes.data), C.c_uint64(n), P(ls.ctypes.data), C.c_uint64(L), C.c_uint64(tile), C.c_uint64(split_idx),
                  P(psig.ctypes.data), P(info.ctypes.data), P(reg.ctypes.data))
    pref = np.ascontiguousarray(o.leaf_params)
    disc = np.zeros(L); closest = np.zeros(L); er = np.zeros(L, dtype=np.uint64); es = np.zeros(L, dtype=np.uint64)
    lib.sigma_eval(P(keys.ctypes.data), C.c_uint64(n), P(ls.ctypes.data), C.c_uint64(L), P(pref.ctypes.data), P(psig.ctypes.data),
                   P(reg.ctypes.data), P(disc.ctypes.data), P(closest.ctypes.data), P(er.ctypes.data), P(es.ctypes.data))
    u = 2.0 ** -53
    cnt, X, W, sg = info[:, 0], info[:, 1], info[:, 2], info[:, 3]
    beta = np.abs(psig[:, 1]); Y = ls[1:].astype(np.float64)
    with np.errstate(all="ignore"):
        bound = (cnt * beta * X + cnt * beta * W * X / sg + beta * X + Y + 4 * info[:, 4] * beta * W) * u
        ratio = np.where(reg.astype(bool), disc / bound, 0)
    idx = np.argsort(-ratio)[:k]
    for j in idx:
        s, e = int(ls[j]), int(ls[j + 1])
        tb = (e // tile) * tile
        D = float(keys[e]) - float(keys[tb])
        print(f"leaf {j}: n={cnt[j]:.0f} ratio={ratio[j]:.3f} disc={disc[j]:.3e} X={X[j]:.3e} W={W[j]:.3e} sg={sg[j]:.3e} W/sg={W[j]/sg[j]:.2f} beta={beta[j]:.3e} Dpivot/sg={D/sg[j]:.1f} "
              f"pref={pref[j]} psig={psig[j]}")

if __name__ == "__main__" and len(sys.argv) > 4:
    top(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))

def quant(gen, n, L, tile=4096):
    keys = datagen.GENERATORS[gen](n)
    o = orc.train_two_layer("linear", "linear", keys, L, threads=2)
    ls = np.ascontiguousarray(o.leaf_start)
    mid = L // 2
    split_idx = int(ls[mid]) if ls[mid] < n else n
    psig = np.zeros((L, 2)); info = np.zeros((L, 5)); reg = np.zeros(L, dtype=np.uint8)
    lib.sigma_fit(P(keys.ctypes.data), C.c_uint64(n), P(ls.ctypes.data), C.c_uint64(L), C.c_uint64(tile), C.c_uint64(split_idx),
                  P(psig.ctypes.data), P(info.ctypes.data), P(reg.ctypes.data))
    pref = np.ascontiguousarray(o.leaf_params)
    r = reg.astype(bool)
    rb = np.abs(psig[r, 1] - pref[r, 1]) / np.abs(pref[r, 1])
    X = info[r, 1]
    scale = np.abs(pref[r, 1]) * X + ls[1:][r]
    ra = np.abs(psig[r, 0] - pref[r, 0]) / scale
    for q in (0.5, 0.9, 0.99, 0.999, 0.9999, 1.0):
        print(f"  q{q}: beta rel {np.quantile(rb, q):.3e}   alpha/scale {np.quantile(ra, q):.3e}")
    print("  frac beta rel > 1e-9:", (rb > 1e-9).mean())

if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "quant":
    quant(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
