#!/bin/bash
# round 4: dry run of bench.py's multi-rank control flow on the one-GPU box (2 and 4 ranks on cuda:0 over gloo; not a measurement):
# rank bookkeeping, timing reduction, single_sequence with frame-pair-owned stacks and broadcasts, the guarded closing barrier
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zh; mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  PSFM_BENCH_DRYRUN_ONE_GPU=1 timeout 900 python bench.py --gpus $n --steps 2 --warmup 1 --single-seq-frames 41 > $O/bench_$n.json 2> $O/bench_$n.err
  echo "rc $? ranks $n"; tail -c 1500 $O/bench_$n.json; echo; tail -5 $O/bench_$n.err
done
