#!/bin/bash
# round 6, call a: the driver's own sequence -- smoke(), `python bench.py --gpus 1 --steps 20 --warmup 5` (the new <4 KB line), the GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "smoke: $SECONDS s"; SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
echo "bench rc=$? in $SECONDS s; line bytes: $(tail -1 gpurun_out/r06_a_bench.json | wc -c)"; tail -1 gpurun_out/r06_a_bench.json
tail -3 gpurun_out/r06_a_bench.err
cp bench_extras.json gpurun_out/r06_a_bench_extras.json 2>/dev/null
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "gpu suite: $SECONDS s"
