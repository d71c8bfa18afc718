#!/bin/bash
# round 5, the evidence of the final sources: profiles of every configuration, then the bench line (the driver's flags and the long protocol)
O=gpurun_out/r5final; mkdir -p $O
for c in "M -" "C5 -" "C5 dups" "C3 -" "C2 -"; do set -- $c; tools/profile_r05.sh $1 $2 > $O/prof_$1_$2.log 2>&1; done
for t in m c5 c5_dups c3 c2; do cp gpurun_out/prof_r05/$t/traffic_r05_$t.json profiles/ 2>/dev/null; done
timeout -k 5 900 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r05_bench_driver_flags.json 2> $O/bench1.err
timeout -k 5 900 python bench.py --no-configs < /dev/null > $O/r05_bench.json 2> $O/bench2.err
tail -c 600 $O/bench1.err; python - <<'PY'
import json
for f in ("r05_bench_driver_flags.json", "r05_bench.json"):
    try:
        d = json.loads(open("gpurun_out/r5final/" + f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "ms", round(d["ms_per_step"], 4), "value %.3e" % d["value"], "frac", round(r["frac"], 3), "kernel", r["kernel"], round(r["kernel_frac"], 3), "traffic_ratio", r.get("traffic_ratio"),
              "measured_peak", r.get("measured_peak"), "frac_of_measured", r.get("frac_of_measured"), "parity", d.get("parity_check", {}).get("coef_bit_identical"))
        for k, v in (d.get("configs") or {}).items():
            e = v.get("exact", v)
            print("   ", k, {x: (round(e[x], 4) if isinstance(e.get(x), float) else e.get(x)) for x in ("ms_per_step", "frac_wall", "kernel", "kernel_us", "pipeline")} if isinstance(e, dict) else e)
    except Exception as ex:
        print(f, "ERR", ex)
PY
