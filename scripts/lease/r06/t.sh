#!/bin/bash
# the two dry-run bench tests (children for the one-sequence figures) + the plain N=1 bench line
O=gpurun_out/r06_t; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sharded.py -q -m gpu -k "bench_" -x 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
PSFM_BENCH_DRYRUN_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --single-seq-frames 41 > $O/bench_2.json 2> $O/bench_2.err
tail -c 3000 $O/bench_2.json; tail -5 $O/bench_2.err
