# round 5, call 55: the shipping library, long runs -- 10 000 launches of the ragged packed batch per cell kind (before the fix: 2 of 10 000 GRUmod launches deviated), front_order_diag 500 x 3 orders
mkdir -p gpurun_out/r05final
(timeout 3000 python tools/dev/pack_repeat.py 10000 2>&1 | cut -c1-250 | tail -6) > gpurun_out/r05final/pack_repeat_10000.txt
(timeout 2400 python tools/dev/front_order_diag.py 500 2>&1 | cut -c1-250 | tail -4) > gpurun_out/r05final/front_order_diag_500.txt
cat gpurun_out/r05final/pack_repeat_10000.txt gpurun_out/r05final/front_order_diag_500.txt
