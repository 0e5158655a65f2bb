#!/bin/bash
# Round-4 final evidence on ONE box: gpu tests (+ parity json), default bench line, config-5 shapes, rocprofv3 stats + PMC
# (tools/profile_gpu.sh), timelines, a short soak of the 16-bit formats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
(SCADE_PARITY_JSON=$O/parity.json timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/gputests.txt
timeout 400 python bench.py > $O/bench_final.log 2>&1
timeout 400 python bench.py --rays 4096 --hyp 40 --no-cpu-baseline --no-image > $O/bench_4096.log 2>&1
timeout 1800 tools/profile_gpu.sh r04 > $O/profile_gpu.log 2>&1
for p in f32 f16x3 bf16 bf16-s8; do
  timeout 200 tools/timeline.sh $p 30 python $R/tools/probe_step.py $p 1024 > /dev/null 2>&1
done
(SOAK_STEPS=6000 SOAK_PRECISIONS=f32,bf16,bf16-s8 timeout 300 python tools/soak_train.py 2>&1 | grep -v amdgpu.ids) > $O/soak.txt
tail -3 $O/gputests.txt; tail -c 600 $O/bench_final.log; ls $R/gpurun_out/prof_r04 | head -40
