#!/bin/bash
# round 5, call zo: the ingest stress after the script's own fix (a wrong-sized FIRST file defines the stack's size), one more fresh batch seed
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python scripts/stress_ingest.py 400 51 2>&1 | tail -1 | cut -c1-300 | tee gpurun_out/r05_zo.txt
timeout 300 python scripts/stress_batch.py 120 61 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/r05_zo.txt
