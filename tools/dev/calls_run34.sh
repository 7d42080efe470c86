# round 5, call 34: repeatability of the packed launches that ship (px3), 1200 runs per cell kind
mkdir -p gpurun_out/r05z
timeout 3000 python tools/dev/pack_repeat.py 1200 > gpurun_out/r05z/repeat_tree_1200.txt 2>&1
cat gpurun_out/r05z/repeat_tree_1200.txt
