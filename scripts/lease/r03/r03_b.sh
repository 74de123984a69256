#!/bin/bash
# round 3, run b: the solver core rewrite (psfm_pc_core.h) + the evaluate-ahead launch chain: parity, then timings
O=gpurun_out/r03_b; mkdir -p $O
export PSFM_WHOLE_SEQ_REPORT=$PWD/$O/whole_seq.jsonl
(time timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; tail -4 $O/t1.log
(time timeout 900 python -m pytest tests/test_gpu_whole_sequence.py -x -q -m gpu -k "configs2 or hard") > $O/t2.log 2>&1; tail -4 $O/t2.log
for w in 3 4; do PSFM_SEQ_WAVES=$w PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/easy_w$w.json 2>$O/easy_w$w.err; cat $O/easy_w$w.json; done
for b in 512 1024; do PSFM_PC_BLOCKS=$b PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_b$b.json 2>$O/hard_b$b.err; cat $O/hard_b$b.json; done
