#!/bin/bash
# own sort in the batch finalize: batch parity tests, then the whole suite on the final sources
O=gpurun_out/r06_y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_sort.py -q -m gpu -x 2>&1 | tail -4 | tee $O/batch_tests.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/suite.txt
