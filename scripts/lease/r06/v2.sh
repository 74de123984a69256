#!/bin/bash
# psfm_sort.hip built with -D flags on the box ("$@" = one flag set per argument), per-pass durations of its kernels inside the headline step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_v2; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -mllvm -amdgpu-atomic-optimizer-strategy=None"
cd /tmp && export TMPDIR=/tmp
for F in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $F -c $R/particle-sfm_amd/csrc/psfm_sort.hip -o $R/particle-sfm_amd/build/psfm_sort.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/particle-sfm_amd/lib/libpsfm_hip.so $R/particle-sfm_amd/build/*.o || exit 1
  rm -rf $O/prof
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/bench.py --steps 10 --warmup 2 --no-extras > $O/bench.json 2> $O/bench.err
  python - "$F" $O <<'P'
import csv, glob, json, sys
f, o = sys.argv[1], sys.argv[2]
t = glob.glob(o + "/prof/*/*kernel_trace.csv")[0]
per = {}
for r in csv.DictReader(open(t)):
    nm = r["Kernel_Name"]
    if "psfm_sort" in nm:
        per.setdefault(nm[:24], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
l = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
print("[%s] finalize %.1f us, parity %s" % (f, l["kernels"]["finalize_avg_us"], (l.get("parity") or {}).get("ids_lengths_equal")))
for nm, v in per.items():
    print("   ", nm, "by pass:", [round(sum(v[p::4]) / len(v[p::4]) / 1e3, 1) for p in range(4)], "sum %.1f" % (sum(v) / (len(v) / 4) / 1e3))
P
done
rm -rf $O/prof
