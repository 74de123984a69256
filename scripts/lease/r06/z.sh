#!/bin/bash
# round 6, call z: the evidence for profiles/ on the final sources -- scripts/profile_round5.sh (kernel stats, separate PMC passes, the files
# bench.py replays, the default bench line) -- then the driver's own sequence: smoke(), `python bench.py --gpus 1 --steps 20 --warmup 5`
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
bash scripts/profile_round5.sh r06_z > gpurun_out/r06_z_profile.log 2>&1
echo "profile_round5: $SECONDS s"; tail -5 gpurun_out/r06_z_profile.log
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_z_bench.json 2> gpurun_out/r06_z_bench.err
echo "bench rc=$? in $SECONDS s; line bytes: $(tail -1 gpurun_out/r06_z_bench.json | wc -c)"; tail -1 gpurun_out/r06_z_bench.json
cp bench_extras.json gpurun_out/r06_z_bench_extras.json 2>/dev/null
for f in traffic_chain_fused.json traffic_chain_persist.json traffic_chain_step.json solver_valu.json batch_pmc.json; do cp profiles/$f gpurun_out/r06_z_$f 2>/dev/null; done
