"""Development aid (GPU box): the model-level aggregates of the GPU result against the oracle's, per pipeline."""
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
from lanes_check import mk
from rmi_amd import datagen as dg
from oracle import binding as orc

orc.build()
for n, L in ((300_000, 4096), (2_000_000, 16384)):
    keys = dg.uniform_u64(n)
    o = orc.train_two_layer("linear", "linear", keys, L)
    for name, env in (("lanes", {"RMI_HIP_REGS": "0"}), ("regs", {"RMI_HIP_REGS": "1"})):
        tr = mk(env)
        tr.set_keys(keys)
        g = tr.train("linear,linear", L)
        print(n, L, name, "pipeline", g.pipeline, "l2", repr(g.model_avg_l2_error), repr(o.model_avg_l2_error), "log2", repr(g.model_avg_log2_error), repr(o.model_avg_log2_error),
              "avg", g.model_avg_error == o.model_avg_error, "max", g.model_max_error == o.model_max_error)
        tr.close()
