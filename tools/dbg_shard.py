import sys; sys.path.insert(0,'.')
import numpy as np
from rmi_amd import datagen as dg, train, sharded
keys = dg.dups_u64(200_000); L=8192; G=8
tr = train.Trainer(keys); root = tr.fit_root("radix", L); full = tr.train_leaves(root, "linear_spline", L)
fe = full.last_layer_max_l1s.copy(); fs = full.leaf_starts.copy(); tr.close()
plans = sharded.Planner(lambda i: keys[i], len(keys), keys.dtype, root, L).plan(G)
for pl in plans:
    t = train.Trainer(np.ascontiguousarray(keys[pl.read_lo:pl.read_hi]))
    res = sharded.run_shard(t, pl, root, "linear_spline")
    e = res.last_layer_max_l1s
    bad = np.nonzero(e != fe[pl.leaf_lo:pl.leaf_hi])[0]
    print(pl.rank, pl.key_lo, pl.key_hi, pl.read_lo, pl.read_hi, "bad leaves:", bad[:10], [(int(e[b]), int(fe[pl.leaf_lo+b])) for b in bad[:5]])
    for b in bad[:3]:
        j = pl.leaf_lo + b; s,e_ = int(fs[j]), int(fs[j+1]); print("   leaf", j, "start", s, "end", e_, "keys around start:", keys[max(0,s-3):s+3])
    t.close()
