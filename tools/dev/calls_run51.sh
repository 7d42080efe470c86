# round 5, call 51: the final tree -- smoke(), the GPU suite, ten minutes of the soak tool (pairs and single batches in turn), the driver's bench command
mkdir -p gpurun_out/r05final
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/r05final/smoke.txt
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r05final/suite2.txt
(timeout 700 python tools/stress.py 600 2>&1 | tail -6) > gpurun_out/r05final/stress.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r05final/bench_default2.json
cat gpurun_out/r05final/smoke.txt gpurun_out/r05final/suite2.txt gpurun_out/r05final/stress.txt; python -c "
import json; d=json.load(open('gpurun_out/r05final/bench_default2.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['host_fed']['value'], d['cpu_baseline']['value'])"
