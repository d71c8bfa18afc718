"""Development aid (GPU box): one small linear_spline training through pipeline 5; prints whether it ran through."""
import sys, numpy as np
sys.path.insert(0, ".")
from rmi_amd import train, datagen as dg
n, L = int(sys.argv[1]), int(sys.argv[2])
keys = dg.GENERATORS[sys.argv[3] if len(sys.argv) > 3 else "uniform_u64"](n)
tr = train.Trainer(keys)
root = tr.fit_root("linear", L)
try:
    g = tr.train_leaves(root, "linear_spline", L)
    print("ran: pipeline", g.pipeline, "max err", g.model_max_error)
except Exception as e:
    print("error", e)
