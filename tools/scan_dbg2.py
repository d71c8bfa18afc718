"""Development aid (GPU box): caller-provided roots through pipeline 5 against pipeline 2 and the oracle; prints the leaves that differ."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from rmi_amd import datagen as dg, train
from oracle import binding as orc
orc.build()
keys = dg.uniform_u64(300_000)
L = 4096
for kind, params in [(0, (0.0, 1e-12, 0.0, 0.0))]:
    model = train.Model(kind, params, (0, 0, 0, 0))
    outs = {}
    for pl in ("3", "2"):
        os.environ["RMI_HIP_PIPELINE"] = pl
        tr = train.Trainer(keys)
        g = tr.train_leaves(model, "linear_spline", L).materialize()
        outs[pl] = g
        print("pipeline", pl, "->", g.pipeline, "long", g.long_leaves)
        tr.close()
    try:
        o = orc.train_two_layer_with_root(model, "linear_spline", keys, L) if hasattr(orc, "train_two_layer_with_root") else None
    except Exception as ex:
        print("oracle:", ex); o = None
    a, b = outs["3"], outs["2"]
    for name, x, y in [("starts", a.leaf_starts, b.leaf_starts), ("alpha", a.leaf_params[:, 0].view(np.uint64), b.leaf_params[:, 0].view(np.uint64)),
                       ("beta", a.leaf_params[:, 1].view(np.uint64), b.leaf_params[:, 1].view(np.uint64)), ("err", a.last_layer_max_l1s, b.last_layer_max_l1s),
                       ("count", a.leaf_counts, b.leaf_counts)]:
        bad = np.flatnonzero(x != y)
        print(name, bad.size, bad[:10])
        for j in bad[:6]:
            print("   leaf", j, "p5", a.leaf_params[j], a.leaf_starts[j], a.leaf_starts[min(j + 1, L - 1)], "p2", b.leaf_params[j], b.leaf_starts[j], "keys", keys[max(0, int(b.leaf_starts[j]) - 1):int(b.leaf_starts[j]) + 2])
