#!/bin/bash
# eighth GPU call: split-operand convolution -- suite, speed; CLI throughput by reader count (complete output)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], d["value"], "Msamples/s", d["ms_per_step"], "ms/step; layer", d["roofline"]["avg_launch_ms"], "ms", d["kernel_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2].replace(".json", ".err")).read()[-400:])
PY
}
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
for c in c2 h256 c5 rle; do
  FFHIP_NO_SPLIT_CONV=1 timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_cv0_$c.json 2> gpurun_out/r02_cv0_$c.err; show "f32 conv   $c" gpurun_out/r02_cv0_$c.json
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-h2d-leg > gpurun_out/r02_cv1_$c.json 2> gpurun_out/r02_cv1_$c.err; show "split conv $c" gpurun_out/r02_cv1_$c.json
done
timeout 900 python tools/dev/fp64_truth.py 16 2500 384 2>&1 | tail -7
timeout 1200 python tools/cli_throughput.py 384 2048,8192 0,1,2,4,6 2>&1 | grep -E "readers [0-9]+:|waiting for the reader|read fast5|files listed" > gpurun_out/r02_cli_throughput.txt; cat gpurun_out/r02_cli_throughput.txt
