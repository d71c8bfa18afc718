import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmi_amd import datagen as dg, train
keys = dg.books_u64(150_000)
tr = train.Trainer(keys)
root = tr.fit_root("radix18", 8192)
print("root", root.ip, root.table[:4], root.table.max(), flush=True)
leaf = sys.argv[1] if len(sys.argv) > 1 else "linear"
r = tr.train_leaves(root, leaf, 8192)
print("ok", r.model_max_error, r.long_leaves, flush=True)
