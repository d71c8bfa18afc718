"""Development aid (GPU box): streamed training (train_from_host) of linear_spline leaves against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from rmi_amd import datagen as dg, train
from oracle import binding as orc
orc.build()
gen, spec, L, n = (sys.argv[1:5] + [None] * 4)[:4] if len(sys.argv) > 4 else ("dups_u32", "radix,linear_spline", 1024, 150_000)
L, n = int(L), int(n)
keys = dg.GENERATORS[gen](n)
tr = train.Trainer()
g = tr.train_from_host(keys, spec, L).materialize()
root, leaf = spec.split(",")
o = orc.train_two_layer(root, leaf, keys, L)
print(gen, spec, L, n, "pipeline", g.pipeline, "starts", np.array_equal(g.leaf_starts, o.leaf_start), "params", np.array_equal(g.leaf_params.view(np.uint64), o.leaf_params.view(np.uint64)),
      "err", np.array_equal(g.last_layer_max_l1s, o.leaf_err), flush=True)
