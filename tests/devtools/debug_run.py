"""One-pass (sufficient statistics) mode against the oracle, with a per-leaf dump of what differs.
Development aid for the GPU box:  python tests/devtools/debug_run.py [gen n L root mode] ..."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rmi_amd import datagen as dg, train
from oracle import binding as orc


def run(gen, n, L, root="linear", mode=1, tile=4096, blocks=512, show=12):
    keys = dg.GENERATORS[gen](n)
    tr = train.Trainer(keys)
    tr.set_fit_mode(mode)
    o_root = orc.fit_root(root, keys, L)
    g_root = tr.fit_root(root, L)
    assert g_root.p == o_root.p, (g_root, o_root)
    o = orc.train_two_layer(root, "linear", keys, L, threads=2)
    tr.set_profile_level(2)
    t0 = time.time()
    g = tr.train_leaves(g_root, "linear", L)
    dt = time.time() - t0
    print(f"== {gen} n={n} L={L} root={root} mode={mode}: used={g.fit_mode_used} exact_leaves={g.exact_leaves} merged={g.merged_leaves} guard={g.guard_leaves} "
          f"long={g.long_leaves} device={g.device_ns/1e6:.3f} ms kernels(ms)={[round(k/1e6,3) for k in g.kernel_ns[:5]]} wall={dt*1e3:.1f} ms")
    ls_ok = np.array_equal(g.leaf_starts, o.leaf_start)
    print("   leaf_starts equal:", ls_ok)
    if not ls_ok:
        bad = np.nonzero(g.leaf_starts != o.leaf_start)[0]
        print("   first differing starts:", [(int(j), int(g.leaf_starts[j]), int(o.leaf_start[j])) for j in bad[:show]], "count", len(bad))
    ge, oe = g.last_layer_max_l1s, o.leaf_err
    bad = np.nonzero(ge != oe)[0]
    print(f"   error ints differing: {len(bad)}   counts equal: {np.array_equal(g.leaf_counts, o.leaf_count)}")
    gp, op = g.leaf_params, o.leaf_params
    with np.errstate(all="ignore"):
        relb = np.abs(gp[:, 1] - op[:, 1]) / np.abs(op[:, 1])
    relb[gp[:, 1] == op[:, 1]] = 0
    ident = (gp == op).all(axis=1)
    print(f"   coefficient rows bit-identical: {ident.sum()} of {L};  beta rel: max {np.nanmax(relb):.3e}  >1e-9: {(relb > 1e-9).sum()}  >1e-6: {(relb > 1e-6).sum()}")
    ls = o.leaf_start
    chunk = -(-n // blocks); chunk = -(-chunk // 16) * 16; chunk = max(chunk, tile)
    def describe(j):
        s, e = int(ls[j]), int(ls[j + 1])
        blk_s, blk_e = s // chunk, (e - 1) // chunk if e > s else s // chunk
        ts, te = (s - blk_s * chunk) // tile, (e - blk_s * chunk) // tile
        return (f"leaf {j}: [{s},{e}) len {e-s} blk {blk_s}/{blk_e} tile {ts}/{te} row {(s - blk_s*chunk) % tile // 16}/{(e - blk_s*chunk) % tile // 16} "
                f"err g={int(ge[j])} o={int(oe[j])} beta g={gp[j,1]:.17g} o={op[j,1]:.17g} alpha g={gp[j,0]:.17g} o={op[j,0]:.17g}")
    for j in bad[:show]:
        print("   ERR ", describe(j))
    worst = np.argsort(-np.nan_to_num(relb, nan=np.inf))[:show]
    for j in worst:
        if relb[j] > 1e-6 or np.isnan(relb[j]):
            print("   COEF", describe(j))
    print(f"   aggregates: max {g.model_max_error}/{o.model_max_error} idx {g.model_max_error_idx}/{o.model_max_error_idx} avg {g.model_avg_error}/{o.model_avg_error}")
    tr.close()
    return len(bad) == 0 and ls_ok


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args:
        cases = [("uniform_u64", 200_000, 1024, "linear", 1), ("uniform_u64", 1_000_000, 4096, "linear", 1),
                 ("uniform_u64", 5_000_000, 16384, "linear", 1), ("books_u64", 1_000_000, 4096, "linear", 1),
                 ("dups_u64", 300_000, 1024, "linear", 1), ("clustered_u64", 300_000, 1024, "linear", 1),
                 ("uniform_u32", 1_000_000, 4096, "linear", 1), ("uniform_f64", 300_000, 1024, "linear", 1),
                 ("uniform_u64", 1_000_000, 4096, "radix", 1), ("uniform_u64", 1_000_000, 4096, "cubic", 1),
                 ("uniform_u64", 1_000_000, 4096, "linear", 2),
                 ("uniform_u64", 200_000_000, 1 << 20, "linear", 1)]
    else:
        cases = [(args[i], int(args[i + 1]), int(args[i + 2]), args[i + 3], int(args[i + 4])) for i in range(0, len(args), 5)]
    ok = True
    for c in cases:
        try:
            ok &= run(*c)
        except Exception as ex:  # keep going: one call on the GPU box is precious
            import traceback; traceback.print_exc(); ok = False
    print("ALL OK" if ok else "SOME FAILED")
