# round 5, call 32: is test_dense_launch_matches_the_one_tile_launches[1-1040] a px6 / px7 failure or a flaky test?  five runs each of the tree (px3), px6, px7
mkdir -p gpurun_out/r05y
cp flappie_amd/libffhip.so /tmp/tree0.so
for v in tree px6 px7 tree px6 px7; do
  [ $v = tree ] && cp /tmp/tree0.so flappie_amd/libffhip.so || cp tools/variants/libffhip_$v.so flappie_amd/libffhip.so
  for rep in 1 2 3; do
    r=$(timeout 300 python -m pytest "tests/test_split_gpu.py::test_dense_launch_matches_the_one_tile_launches" -m gpu -q 2>&1 | tail -1)
    echo "$v rep $rep: $r"
  done
done > gpurun_out/r05y/flaky.txt 2>&1
cp tools/variants/libffhip_px6.so flappie_amd/libffhip.so
timeout 300 python -m pytest "tests/test_split_gpu.py::test_dense_launch_matches_the_one_tile_launches" -m gpu -q -x 2>&1 | tail -30 >> gpurun_out/r05y/flaky.txt
cp /tmp/tree0.so flappie_amd/libffhip.so
cat gpurun_out/r05y/flaky.txt
