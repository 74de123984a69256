#!/bin/bash
# PMC passes for the chain kernel (separate passes, --kernel-trace only; see MI355X_MICROARCH.md HBM section)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --frames 31"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM -f csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
ls -R $OUT | head -30
