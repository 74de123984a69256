#!/bin/bash
# round 4: the stage disk to disk -- track.npy with the records built under the write of the point array, and the .flo ingest with 4 / 8 / 16
# reader threads (staging 8 / 16 / 32 frames)
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zg; mkdir -p $O
cd $GRAFT_REPO_ROOT
for k in "4 8" "8 16" "16 32" "4 8"; do
  set -- $k
  echo "readers $1 staging $2" | tee -a $O/e2e.txt
  PSFM_FLO_READERS=$1 PSFM_FLO_STAGING=$2 timeout 300 python scripts/end_to_end.py 101 /dev/shm/psfm_e2e 2> /dev/null | tail -4 | cut -c1-330 | tee -a $O/e2e.txt
done
