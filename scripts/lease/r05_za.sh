#!/bin/bash
# round 5, call za: several host threads on one GPU, random sequences, five ways of setting the workers up (scripts/stress_threads.py)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python scripts/stress_threads.py 25 1 > gpurun_out/r05_za_stress_threads.txt 2>&1; echo "rc=$?" >> gpurun_out/r05_za_stress_threads.txt
grep -v amdgpu.ids gpurun_out/r05_za_stress_threads.txt | tail -12 | cut -c1-400
