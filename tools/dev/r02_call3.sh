#!/bin/bash
# third GPU call: ablation of the fp16 split layer kernel (where do a step's cycles go now), full GPU suite, c5 on the fused N = 4 kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp flappie_amd/libffhip.so /tmp/libffhip_base.so
for tag in a0 a2 a4 a6 a1 a3 a10 a15 a34 a38 a0; do
  cp tools/bin/libffhip_$tag.so flappie_amd/libffhip.so
  timeout 200 python tools/dev/ablate.py $tag 2>&1 | tail -1
done | tee gpurun_out/r02_f16_ablate.txt
cp /tmp/libffhip_base.so flappie_amd/libffhip.so
timeout 600 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r02_f16_bench_c5_fused.json 2> gpurun_out/r02_f16_bench_c5_fused.err
python -c "
import json; d=json.load(open('gpurun_out/r02_f16_bench_c5_fused.json')); print('c5', d['value'], d['ms_per_step'], d['roofline']['kernel'][:30], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | tail -60 > gpurun_out/r02_call3_pytest.txt
grep -E "passed|failed|fuzz tail|GPU (split|f32)|oracle \(reference|trace cells" gpurun_out/r02_call3_pytest.txt
