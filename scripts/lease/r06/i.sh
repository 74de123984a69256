#!/bin/bash
# round 6, call i: ingest probe (copy streams x readers x staging)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cs in 1 2; do for rd in 8 16 24; do
  PSFM_FLO_COPY_STREAMS=$cs PSFM_FLO_READERS=$rd PSFM_FLO_STAGING=$((2*rd)) timeout 200 python scripts/probe_ingest.py 100 keep 2>&1 | tail -1
done; done | tee gpurun_out/r06_i_ingest.txt
rm -rf /dev/shm/psfm_ingest_probe
