#!/bin/bash
# rocprofv3 evidence for profiles/ (round 5) = scripts/profile_round4.sh (the single-sequence kernels: kernel-trace stats and SEPARATE
# PMC passes of the headline step, the track_optimize path on clean and hard flows -- which refreshes what bench.py replays for the
# sources at hand) + the BATCHED launches (psfm_connect_batch: DAVIS x 16 track, Sintel x 16 optimize) + realistic flows.
# scripts/summarize_profiles5.py boils it down on the box.  Never --pmc together with other trace domains.
# Usage (on the GPU box): bash scripts/profile_round5.sh r05_x      -> gpurun_out/r05_x_summary/
TAG=${1:-r05}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
B1="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-extras"
B2="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras"
PR="python $GRAFT_REPO_ROOT/scripts/probe_solver.py"
RB="python $GRAFT_REPO_ROOT/scripts/run_batch_once.py"
run() { d=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" > $OUT/$d.log 2>&1 < /dev/null; }
RD="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum"
WR="TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum TCC_EA0_WRREQ_sum"
# ---- headline step ----
run stats --stats -f csv -d $OUT/stats -o $TAG -- $B1
run fused_fetch --pmc $RD -f csv -d $OUT/fused_fetch -o f -- $B2
run fused_write --pmc $WR -f csv -d $OUT/fused_write -o w -- $B2
PSFM_BENCH_TWO_CALLS=1 run two_fetch --pmc $RD -f csv -d $OUT/two_fetch -o f -- $B2
PSFM_BENCH_TWO_CALLS=1 run two_write --pmc $WR -f csv -d $OUT/two_write -o w -- $B2
PSFM_BENCH_CHAIN_MODE=1 run step_fetch --pmc $RD -f csv -d $OUT/step_fetch -o f -- $B2
PSFM_BENCH_CHAIN_MODE=1 run step_write --pmc $WR -f csv -d $OUT/step_write -o w -- $B2
# ---- track_optimize on clean flows (the device-paced frame kernel) ----
export PSFM_PROBE_MODES=adaptive
run opt_stats --stats -f csv -d $OUT/opt_stats -o ${TAG}_opt -- $PR
run opt_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-include-regex "psfm_" -f csv -d $OUT/opt_pmc_sq -o s -- $PR
run opt_fetch --pmc $RD --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_fetch -o f -- $PR
run opt_write --pmc $WR --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_write -o w -- $PR
for SH in "436 1024 50 2" "480 640 200 1"; do
  T=$(echo $SH | tr ' ' 'x')
  run opt_fetch_$T --pmc $RD --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_fetch_$T -o f -- $PR $SH
  run opt_write_$T --pmc $WR --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_write_$T -o w -- $PR $SH
  run opt_pmc_sq_$T --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-include-regex "psfm_seq" -f csv -d $OUT/opt_pmc_sq_$T -o s -- $PR $SH
done
# ---- hard flows (sigma 0.3, 5 % occluders): the resident solve.  SQ counters only: TA / TCC passes hang kernels with a device-wide hand-off ----
export PSFM_PROBE_HARD=1
run hard_stats --stats -f csv -d $OUT/hard_stats -o ${TAG}_hard -- $PR
run hard_pmc_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-include-regex "psfm_pc_" -f csv -d $OUT/hard_pmc_sq -o s -- $PR
unset PSFM_PROBE_HARD PSFM_PROBE_MODES
# ---- B sequences per launch ----
run batch_davis_stats --stats -f csv -d $OUT/batch_davis_stats -o ${TAG}_bd -- $RB davis 16 5
run batch_davis_fetch --pmc $RD --kernel-include-regex "psfm_chain_step_batch" -f csv -d $OUT/batch_davis_fetch -o f -- $RB davis 16 2
run batch_davis_write --pmc $WR --kernel-include-regex "psfm_chain_step_batch" -f csv -d $OUT/batch_davis_write -o w -- $RB davis 16 2
run batch_sintel_stats --stats -f csv -d $OUT/batch_sintel_stats -o ${TAG}_bs -- $RB sintel 16 5
run batch_sintel_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-include-regex "psfm_seq_batch" -f csv -d $OUT/batch_sintel_pmc_sq -o s -- $RB sintel 16 2
run batch_scannet_stats --stats -f csv -d $OUT/batch_scannet_stats -o ${TAG}_bn -- $RB scannet 4 3
run batch_scannet_pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-include-regex "psfm_seq_batch" -f csv -d $OUT/batch_scannet_pmc_sq -o s -- $RB scannet 4 2
# ---- boil it down, put the files bench.py replays (stamped with this binary's source hash) in place ON THE BOX, then the default bench
#      line of this binary with them (`same_sources: true`); the same files are committed under profiles/ afterwards ----
python $GRAFT_REPO_ROOT/scripts/summarize_profiles5.py $TAG
SUM=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_summary
for f in traffic_chain_fused.json traffic_chain_persist.json traffic_chain_step.json solver_valu.json batch_pmc.json; do
  [ -f $SUM/$f ] && cp $SUM/$f $GRAFT_REPO_ROOT/profiles/$f
done
timeout 1200 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $SUM/${TAG}_bench.json 2> $SUM/${TAG}_bench.err < /dev/null
ls $SUM
