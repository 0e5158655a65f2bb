#!/bin/bash
# Final evidence of a round on ONE box (tools/final_evidence.sh <tag>, default r05): gpu tests (+ parity json), default bench line, config-5 shapes, rocprofv3 stats + PMC
# (tools/profile_gpu.sh), timelines, a short soak of the 16-bit formats.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
(SCADE_PARITY_JSON=$O/parity.json SCADE_BUCKET_JSON=$O/bucket_grads.json timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/gputests.txt
timeout 400 python bench.py > $O/bench_final.log 2>&1
timeout 400 python bench.py --rays 4096 --hyp 40 --no-cpu-baseline --no-image > $O/bench_4096.log 2>&1
timeout 1800 tools/profile_gpu.sh $TAG > $O/profile_gpu.log 2>&1
for p in f32 f16x3 bf16 bf16-s8; do
  timeout 200 tools/timeline.sh $p 30 python $R/tools/probe_step.py $p 1024 > /dev/null 2>&1
done
timeout 200 tools/timeline.sh f32_128 30 python $R/tools/probe_step.py f32 128 graph > /dev/null 2>&1
timeout 200 tools/timeline.sh bf16-s8_128 30 python $R/tools/probe_step.py bf16-s8 128 graph > /dev/null 2>&1
# the driver's loop (graph-captured step + fused batch gather) against the eager loop, and the fused tail kernel
(ITERS=300 timeout 300 python tools/probe_driver.py f32 bf16-s8 2>&1 | grep -v amdgpu.ids) > $O/driver_loop.txt
(for sz in 128:20 1024:20 512:40 4096:40; do bash tools/kstats.sh tail_$sz python $R/tools/probe_tail_train.py $sz | grep -i tail_train; done) > $O/tail_train.txt 2>&1
(SOAK_STEPS=6000 SOAK_PRECISIONS=f32,f16x3,bf16,bf16-s8 timeout 300 python tools/soak_train.py 2>&1 | grep -v amdgpu.ids) > $O/soak.txt
# the multi-rank logic on one GPU (8 ranks, gloo, shared device: not a measurement), trained-quality parity with its noise floor,
# the per-tensor gradient comparison between precisions, the k-loop probe
(SCADE_BENCH_SHARE_GPU=1 SCADE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --secondary-budget 800 2> $O/bench_8rank_gloo.err | grep '^{"metric' | tail -1) > $O/bench_8rank_gloo.json
timeout 1200 python tools/convergence_parity.py --iters 5000 --seeds 5 --out $O/convergence.json > $O/convergence.log 2>&1
PRECS=f32,f16x3,bf16,bf16-s8 timeout 600 python tools/probe_bucket.py 1024 3072 4096 > $O/probe_bucket.txt 2>&1
(cd /tmp; for M in 0 1 2 3 4 5; do timeout 60 $R/tools/scratch/probe_kloop_wlds $M; done) > $O/kloop.txt 2>&1
tail -3 $O/gputests.txt; tail -c 600 $O/bench_final.log; ls $R/gpurun_out/prof_$TAG | head -40
