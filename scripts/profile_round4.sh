#!/bin/bash
# rocprofv3 evidence for profiles/ (round 4): kernel-trace stats and SEPARATE PMC passes (never --pmc together with other trace
# domains) of the headline step (fused / two-call / per-frame), the track_optimize path on clean flows (frame kernel) and on hard
# flows (resident solve), then the default bench line of the same binary.  scripts/summarize_profiles4.py boils it down on the box
# (kernel stats, PMC summaries, profiles/traffic_chain_*.json and profiles/solver_valu.json stamped with the source hash).
# Usage (on the GPU box): bash scripts/profile_round4.sh r04_x      -> gpurun_out/r04_x_summary/
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
B1="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extras"
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
PR="python $GRAFT_REPO_ROOT/scripts/probe_solver.py"
run() { d=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" > $OUT/$d.log 2>&1 < /dev/null; }
# ---- headline step ----
run stats --stats -f csv -d $OUT/stats -o $TAG -- $B1
# HBM-side bytes: the L2s' fabric request counters in 32-byte units (a 128-byte request counts 4) -- exact on the stand-alone flow_check
# launch (3.330 GB counted for 3.318 GB read, 207.4 MB for 207.4 MB written), unlike FETCH_SIZE, which tallies this chip's 128-byte
# requests at 64 bytes (MI355X_MICROARCH.md: "reports exactly 1/2 of a wide coalesced read"; rounds 2-3 calibrated it on flow_check)
RD="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum"
WR="TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum TCC_EA0_WRREQ_sum"
run fused_fetch --pmc $RD -f csv -d $OUT/fused_fetch -o f -- $B2
run fused_write --pmc $WR -f csv -d $OUT/fused_write -o w -- $B2
PSFM_BENCH_TWO_CALLS=1 run two_fetch --pmc $RD -f csv -d $OUT/two_fetch -o f -- $B2
PSFM_BENCH_TWO_CALLS=1 run two_write --pmc $WR -f csv -d $OUT/two_write -o w -- $B2
PSFM_BENCH_CHAIN_MODE=1 run step_fetch --pmc $RD -f csv -d $OUT/step_fetch -o f -- $B2
PSFM_BENCH_CHAIN_MODE=1 run step_write --pmc $WR -f csv -d $OUT/step_write -o w -- $B2
run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -f csv -d $OUT/pmc_sq -o s -- $B2
# ---- track_optimize on clean flows (the device-paced frame kernel) ----
export PSFM_PROBE_MODES=adaptive
run opt_stats --stats -f csv -d $OUT/opt_stats -o ${TAG}_opt -- $PR
run opt_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-include-regex "psfm_" -f csv -d $OUT/opt_pmc_sq -o s -- $PR
# (the frame kernel has no device-wide hand-off -- its last block to arrive does the control step -- so the L2 counters are safe on it)
run opt_fetch --pmc $RD --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_fetch -o f -- $PR
run opt_write --pmc $WR --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_write -o w -- $PR
# ... and on the other two shapes bench.py reports a frame-kernel roofline for (configs[2]: 436 x 1024, configs[4]: dense 480 x 640)
for SH in "436 1024 50 2" "480 640 200 1"; do
  T=$(echo $SH | tr ' ' 'x')
  run opt_fetch_$T --pmc $RD --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_fetch_$T -o f -- $PR $SH
  run opt_write_$T --pmc $WR --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_write_$T -o w -- $PR $SH
done
# ---- hard flows (sigma 0.3, 5 % occluders): the resident solve.  SQ counters only: TA / TCC passes hang kernels with a device-wide hand-off ----
export PSFM_PROBE_HARD=1
run hard_stats --stats -f csv -d $OUT/hard_stats -o ${TAG}_hard -- $PR
run hard_pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-include-regex "psfm_pc_" -f csv -d $OUT/hard_pmc_sq -o s -- $PR
unset PSFM_PROBE_HARD PSFM_PROBE_MODES
# ---- the default bench line of this binary ----
timeout 900 python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err < /dev/null
python $GRAFT_REPO_ROOT/scripts/summarize_profiles4.py $TAG
ls $GRAFT_REPO_ROOT/gpurun_out/${TAG}_summary
