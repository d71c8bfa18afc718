#!/bin/bash
# the two bench lines of the evidence (the driver's flags with the side configurations, the long protocol) -> gpurun_out/r5final/
O=gpurun_out/r5final; mkdir -p $O
timeout -k 5 600 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r05_bench_driver_flags.json 2> $O/bench1.err
timeout -k 5 600 python bench.py --no-configs < /dev/null > $O/r05_bench.json 2> $O/bench2.err
tail -c 300 $O/bench1.err; python - <<'PY'
import json
for f in ("r05_bench_driver_flags.json", "r05_bench.json"):
    try:
        d = json.loads(open("gpurun_out/r5final/" + f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms", round(d["ms_per_step"], 4), "value %.3e" % d["value"], "frac", round(r["frac"], 3), "kernel", r["kernel"], round(r["kernel_frac"], 3), "traffic_ratio", r.get("traffic_ratio"),
              "measured_peak", r.get("measured_peak"), "frac_of_measured", r.get("frac_of_measured"), "parity", d.get("parity_check", {}).get("coef_bit_identical"))
        for k, v in (d.get("configs") or {}).items():
            e = v.get("exact", v)
            print("   ", k[:28], {x: (round(e[x], 4) if isinstance(e.get(x), float) else e.get(x)) for x in ("ms_per_step", "frac_wall", "kernel", "kernel_us", "pipeline", "traffic_ratio")} if isinstance(e, dict) else e)
    except Exception as ex:
        print(f, "ERR", ex)
PY
