# round 5, call 48: the tree with LDS flag words as the default -- whole GPU suite, the re-sweep test, repeatability of the packed launches (3000 per kind), front_order_diag (all orders),
# the soak tool, then every config's bench line
mkdir -p gpurun_out/r05z gpurun_out/r05final
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r05final/suite.txt
(timeout 1500 python tools/dev/pack_repeat.py 3000 2>&1 | cut -c1-250 | tail -6) > gpurun_out/r05final/pack_repeat.txt
(timeout 1500 python tools/dev/front_order_diag.py 300 2>&1 | cut -c1-250 | tail -4) > gpurun_out/r05final/front_order_diag.txt
for c in c2 h256 c4 c5 rle; do timeout 900 python bench.py --config $c --no-host-fed-leg 2>gpurun_out/r05final/$c.err | tail -1 > gpurun_out/r05final/${c}_bench.json; done
cat gpurun_out/r05final/suite.txt gpurun_out/r05final/pack_repeat.txt gpurun_out/r05final/front_order_diag.txt
python - <<'PY'
import json
for c in ("c2","h256","c4","c5","rle"):
    try:
        d=json.load(open("gpurun_out/r05final/%s_bench.json"%c)); print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("exposed_ms"), d.get("decode_hbm",{}).get("achieved"), d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(c, "failed", e)
PY
