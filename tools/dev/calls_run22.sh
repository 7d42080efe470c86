# round 5, call 22: front_order=head (the next batch's convolutions start behind the previous batches' CRF heads) against the default, every config
mkdir -p gpurun_out/r05o
CFGS="h256 c4 c2 rle" REPS=2 STEPS=30 tools/dev/ab/debug_ab.sh "" "front_order=head" > gpurun_out/r05o/ab.txt 2>&1
cat gpurun_out/r05o/ab.txt
