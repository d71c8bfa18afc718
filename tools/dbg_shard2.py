import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from rmi_amd import datagen as dg, train as T, sharded
n=int(sys.argv[1]) if len(sys.argv)>1 else 100_000_000; L=524288; G=2
tr = T.Trainer(); tr.generate_keys("uniform", np.uint64, n)
root = tr.fit_root("linear", L)
full = tr.train_leaves(root, "linear", L); frows = full.rows.copy(); tr.close()
print("full ok", flush=True)
f = dg.uniform_u64
plans = sharded.Planner(lambda i: f(n, start=i, count=1)[0], n, np.uint64, root, L).plan(G)
buf = torch.zeros(L*24, dtype=torch.uint8, device="cuda")
for pl in plans:
    print(pl, flush=True)
    t = T.Trainer(); t.generate_keys("uniform", np.uint64, n, pl.read_lo, pl.read_hi-pl.read_lo)
    for ext in (False, True):
        res = sharded.run_shard(t, pl, root, "linear", buf.data_ptr()+pl.leaf_lo*24 if ext else None)
        torch.cuda.synchronize()
        print("  shard", pl.rank, "ext", ext, "ok", flush=True)
    t.close()
print("rows equal:", np.array_equal(buf.cpu().numpy(), frows))
a = buf.cpu().numpy().view(np.uint64).reshape(L,3); b = frows.view(np.uint64).reshape(L,3)
bad = np.nonzero((a!=b).any(axis=1))[0]
print("bad rows", len(bad), bad[:10], bad[-5:])
for j in bad[:5]: print(j, a[j].view(np.float64)[:2], a[j][2], b[j].view(np.float64)[:2], b[j][2])
