# round 5, call 54: the x waves' poll of the "consumed" flag (LDS accesses now): s_sleep 0 / 1 / 3 / 7 between polls
mkdir -p gpurun_out/r05final
CFGS="c2 h256" REPS=3 STEPS=40 tools/dev/ab/multi_ab.sh xs1 xs0 xs3 xs7 > gpurun_out/r05final/xpoll_ab.txt 2>&1
cat gpurun_out/r05final/xpoll_ab.txt
