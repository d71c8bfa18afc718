python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from rmi_amd import train as T
tr=T.Trainer(); tr.generate_keys("uniform", np.uint64, 200_000_000)
print("read BW GB/s:", tr.measure_read_bandwidth(10), tr.measure_read_bandwidth(10))
PY
for d in 0 1 2 3; do echo "err_wave dbg=$d"; RMI_HIP_DBG=$d python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_us'])"; done
