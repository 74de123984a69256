#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3): the headline bench step AND the track_optimize path (fused solve).
# kernel-trace stats first, then separate PMC passes (never --pmc together with other trace domains).
# Usage (on the GPU box): bash scripts/profile_round2.sh r03_a      -> gpurun_out/r03_a/...
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
B1="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extras"
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
export PSFM_PROBE_MODES=fused
O1="python $GRAFT_REPO_ROOT/scripts/probe_solver.py"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o $TAG -- $B1 > $OUT/bench_under_rocprof.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o f -- $B2 > $OUT/pmc_fetch.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/pmc_write -o w -- $B2 > $OUT/pmc_write.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -f csv -d $OUT/pmc_sq -o s -- $B2 > $OUT/pmc_sq.log 2>&1 < /dev/null
if [ -z "$PSFM_PROFILE_HEADLINE_ONLY" ]; then
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/opt_stats -o ${TAG}_opt -- $O1 > $OUT/opt_under_rocprof.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/opt_pmc_fetch -o f -- $O1 > $OUT/opt_pmc_fetch.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/opt_pmc_write -o w -- $O1 > $OUT/opt_pmc_write.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -f csv -d $OUT/opt_pmc_sq -o s -- $O1 > $OUT/opt_pmc_sq.log 2>&1 < /dev/null
# the hard distribution (sigma 0.3, 5 % occluders): every solve through the launch chain / the persistent solve
timeout 300 env PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive rocprofv3 --kernel-trace --stats -f csv -d $OUT/hard_stats -o ${TAG}_hard -- python $GRAFT_REPO_ROOT/scripts/probe_solver.py > $OUT/hard_under_rocprof.log 2>&1 < /dev/null
fi   # PSFM_PROFILE_HEADLINE_ONLY=1: only the headline step (the track_optimize kernels did not change)
timeout 500 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err < /dev/null



python $GRAFT_REPO_ROOT/scripts/summarize_profiles3.py $TAG
ls -la $GRAFT_REPO_ROOT/gpurun_out/${TAG}_summary
