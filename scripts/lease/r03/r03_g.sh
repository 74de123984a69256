#!/bin/bash
# round 3, run g: the launch chain's loop as one persistent launch -- small parity first (under a short timeout), then timing
O=gpurun_out/r03_g; mkdir -p $O
(time timeout 240 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "persistent_launch or optimize_location") > $O/t0.log 2>&1; tail -4 $O/t0.log
if grep -q "passed" $O/t0.log && ! grep -q "failed" $O/t0.log; then
  (time timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py -x -q -m gpu) > $O/t1.log 2>&1; tail -4 $O/t1.log
  for p in 1 0; do PSFM_PC_PERSIST=$p PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_persist$p.json 2>$O/hard_persist$p.err; cat $O/hard_persist$p.json; done
  for b in 256 1024; do PSFM_PC_BLOCKS=$b PSFM_PROBE_HARD=1 PSFM_PROBE_MODES=adaptive timeout 300 python scripts/probe_solver.py > $O/hard_persist_b$b.json 2>$O/hard_persist_b$b.err; cat $O/hard_persist_b$b.json; done
  export PSFM_WHOLE_SEQ_REPORT=$PWD/$O/whole_seq.jsonl
  (time timeout 900 python -m pytest tests/test_gpu_whole_sequence.py -x -q -m gpu -k "hard or largemotion") > $O/t2.log 2>&1; tail -4 $O/t2.log
fi
