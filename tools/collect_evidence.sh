#!/bin/bash
# After `gpurun -- 'bash tools/final_evidence.sh <tag>'`: copy what the run left under gpurun_out/ into profiles/ under the
# round's names, run the consistency gate (tools/update_profiles.py) and regenerate DESIGN.md's status table.
#   bash tools/collect_evidence.sh r06
set -e
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd); F=$R/gpurun_out/final; P=$R/profiles
cd $R
python tools/update_profiles.py $TAG gpurun_out/final/bench_final.log
grep '^{"metric' $F/bench_4096.log | tail -1 > $P/${TAG}_bench_line_4096rays_k40.json
tail -8 $F/gputests.txt > $P/${TAG}_gputests.txt
cp $F/parity.json $P/${TAG}_parity.json
cp $F/driver_loop.txt $P/${TAG}_driver_loop.txt
cp $F/tail_train.txt $P/${TAG}_ray_tail_train.txt
cp $F/soak.txt $P/${TAG}_soak.txt
cp $F/convergence.json $P/${TAG}_convergence.json
cp $F/bucket_grads.json $P/${TAG}_bucket_grads.json
cp $F/bench_8rank_gloo.json $P/${TAG}_bench_line_8rank_gloo_shared_gpu_logic_test.json
for p in f32 f16x3 bf16 bf16-s8; do cp gpurun_out/timeline_$p.txt $P/${TAG}_train_step_timeline_$p.txt; done
cp gpurun_out/timeline_f32_128.txt $P/${TAG}_train_step_timeline_f32_128rays_graph.txt
cp gpurun_out/timeline_bf16-s8_128.txt $P/${TAG}_train_step_timeline_bf16-s8_128rays_graph.txt
python tools/design_status.py $TAG
ls -la $P | grep ${TAG}_ | wc -l
