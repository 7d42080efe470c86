# round 5, call 49: the final tree's profile set -- c2 (bench line, kernel stats, HBM and SQ counter passes), h256 and c4 kernel stats, and the driver's default command with its kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
tools/profile_config.sh r05_c2 c2 > /dev/null 2>&1
PROFILE_ONLY=1 true
for c in h256 c4; do
  rm -rf /tmp/ks_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python bench.py --config $c --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg > /dev/null 2>&1
  f=$(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1); (echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --config $c --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-h2d-leg --no-host-fed-leg"; head -16 $f) > gpurun_out/profiles/r05_${c}_kernel_stats.csv
done
timeout 900 python bench.py 2>gpurun_out/profiles/default.err | tail -1 > gpurun_out/profiles/r05_bench_default.json
rm -rf /tmp/ks_def; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_def -- python bench.py --no-host-fed-leg --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/ks_def -name "*kernel_stats.csv" | head -1); (echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-host-fed-leg --no-cpu-baseline   (the driver's command, two pairs in flight)"; head -16 $f) > gpurun_out/profiles/r05_bench_default_kernel_stats.csv
ls gpurun_out/profiles/; python - <<'PY'
import json
d=json.load(open("gpurun_out/profiles/r05_bench_default.json")); print("default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("exposed_ms"), "host_fed", d.get("host_fed",{}).get("value"), d.get("host_fed",{}).get("rank0_walls_s"), "cpu", d.get("cpu_baseline",{}).get("value"), "h2d", d.get("h2d_inclusive",{}).get("value"))
PY
head -5 gpurun_out/profiles/r05_bench_default_kernel_stats.csv | cut -c1-160
