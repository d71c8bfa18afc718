"""Read-only streaming rate of this box over a 1.6 GB resident array (rmi_hip_measure_read_bandwidth)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rmi_amd import train
tr = train.Trainer()
tr.generate_keys("uniform", np.uint64, 200_000_000)
print("read bandwidth GB/s:", [round(tr.measure_read_bandwidth(10)) for _ in range(3)])
