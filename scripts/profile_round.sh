#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the default bench command, then separate PMC passes
# (no --pmc together with other trace domains; see MI355X_MICROARCH.md).  Usage (on the GPU box): bash scripts/profile_round.sh r01_b
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extras > $OUT/bench_under_rocprof.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o f -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_write -o w -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -f csv -d $OUT/pmc_sq -o s -- $CMD > $OUT/pmc_sq.log 2>&1
python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -40
