#!/bin/bash
# round-6 first evidence call: gpu tests (new full-size bucket gradients), 8-rank gloo logic run of bench.py on one GPU, convergence parity
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
(SCADE_BUCKET_JSON=$O/bucket_grads.json timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/gputests.txt
(SCADE_BENCH_SHARE_GPU=1 SCADE_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 1 --secondary-budget 1200 > $O/bench_8rank_gloo.json 2> $O/bench_8rank_gloo.err)
echo "8rank rc=$?" >> $O/gputests.txt
(timeout 1500 python tools/convergence_parity.py --iters 5000 --seeds 5 --out $O/convergence.json > $O/convergence.log 2>&1)
tail -5 $O/gputests.txt; tail -c 400 $O/bench_8rank_gloo.json; tail -5 $O/convergence.log
