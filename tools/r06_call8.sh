#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O
cd /tmp
for rep in 1 2; do for M in 0 1 2 3 4 5; do timeout 60 $R/tools/scratch/probe_kloop_wlds $M; done; done > $O/kloop.txt 2>&1
cd $R; timeout 500 bash tools/pmc_kloop_wlds.sh 2>&1 | grep -v "^mode" > $O/kloop_pmc.txt
cat $O/kloop.txt $O/kloop_pmc.txt | cut -c1-250
