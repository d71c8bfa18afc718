"""Development aid (GPU box): train_streamed with a given number of chunks against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
import torch
torch.cuda.init()
from rmi_amd import datagen as dg, train
from oracle import binding as orc
orc.build()
gen, rootn, L, n, chunks = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
keys = dg.GENERATORS[gen](n)
tr = train.Trainer()
root = tr.fit_root_host(keys, rootn, L)
g = tr.train_streamed(keys, root, "linear_spline", L, chunks=chunks).materialize()
o = orc.train_two_layer(rootn, "linear_spline", keys, L)
bad = np.flatnonzero((g.leaf_params.view(np.uint64) != o.leaf_params.view(np.uint64)).any(axis=1))
bade = np.flatnonzero(g.last_layer_max_l1s != o.leaf_err)
print(gen, rootn, L, n, "chunks", chunks, "pipeline", g.pipeline, "starts", np.array_equal(g.leaf_starts, o.leaf_start), "params bad", bad.size, bad[:8], "err bad", bade.size, bade[:8], flush=True)
if bad.size:
    j = int(bad[0]); print("  leaf", j, "gpu", g.leaf_params[j], "oracle", o.leaf_params[j], "start", o.leaf_start[j], o.leaf_start[min(j+1, L-1)])
