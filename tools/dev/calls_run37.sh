# round 5, call 37: k_grumod_pack with the projection partials released BEHIND the recurrent pass again (pr3 = px3 + late release: the cure of the one-tile-wrong failure) against px3 and px0
mkdir -p gpurun_out/r05z
CFGS="c4" REPS=3 STEPS=30 tools/dev/ab/multi_ab.sh px0 px3 pr3 > gpurun_out/r05z/ab_pr3.txt 2>&1
cat gpurun_out/r05z/ab_pr3.txt
