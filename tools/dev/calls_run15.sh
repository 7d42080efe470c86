mkdir -p gpurun_out/r05h gpurun_out/profiles
python tools/parity_modes.py gpu tools/variants/parity_modes_oracle.npz > gpurun_out/r05h/parity_modes.txt 2>&1
tools/profile_config.sh r05_c2 c2 > gpurun_out/r05h/profile_c2.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_default -- python bench.py --no-cpu-baseline --no-host-fed-leg > gpurun_out/r05h/bench_default_under_rocprof.json 2> gpurun_out/r05h/bench_default.err
f=$(find gpurun_out/prof_r05_default -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/profiles/r05_bench_default_kernel_stats.csv
cat gpurun_out/r05h/parity_modes.txt; head -4 gpurun_out/profiles/r05_bench_default_kernel_stats.csv | cut -c1-200; ls gpurun_out/profiles
