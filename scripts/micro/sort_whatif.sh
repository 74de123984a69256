#!/bin/bash
# timing builds of csrc/psfm_sort.hip with parts of its kernels removed (PS_WHATIF), per-kernel durations by rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sort_whatif; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -mllvm -amdgpu-atomic-optimizer-strategy=None"
cd /tmp && export TMPDIR=/tmp
for W in ${@:-0 1 8 4 6 16}; do      # (2 alone is not safe: it must come with 4)
  /opt/rocm/bin/hipcc $FLAGS $SORT_FLAGS -DPS_WHATIF=$W -c $R/particle-sfm_amd/csrc/psfm_sort.hip -o $R/particle-sfm_amd/build/psfm_sort.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/particle-sfm_amd/lib/libpsfm_hip.so $R/particle-sfm_amd/build/*.o || exit 1
  rm -rf $O/prof
  timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/scripts/micro/sort_probe.py 2073277 32 20 $([ $W = 0 ] && echo 1 || echo 0) > $O/run_$W.txt 2>&1
  python - $W $O <<'P'
import csv, glob, sys
w, o = sys.argv[1], sys.argv[2]
f = glob.glob(o + "/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "psfm_sort" in r["Name"]:
        print("whatif %2s  %-28s calls %4s avg %8.1f ns" % (w, r["Name"][:28], r["Calls"], float(r["AverageNs"])))
P
  tail -1 $O/run_$W.txt
done
rm -rf $O/prof
