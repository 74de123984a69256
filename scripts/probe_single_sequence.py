#!/usr/bin/env python3
"""bench.py's single_sequence figure alone (ONE sequence through psfm_dist.connect_sharded at world size 1, beside the
one-GPU call): for A/B runs of the sharded engine under environment knobs (PSFM_FUSED_WAVES=3|4 ...).

    python scripts/probe_single_sequence.py [frames=201]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import torch
import bench_common as bench
import bench_extras

import gc
if os.environ.get("PSFM_PROBE_GC") == "freeze":      # everything imported so far out of the collector's way (what bench.py does)
    gc.collect()
    gc.freeze()
elif os.environ.get("PSFM_PROBE_GC") == "log":
    import time as _t
    _g = {}
    def _cb(phase, info):
        if phase == "start":
            _g["t"] = _t.perf_counter()
        elif info["generation"] == 2:
            print("[gc] generation 2: %.1f ms at t=%.3f" % (1e3 * (_t.perf_counter() - _g["t"]), _t.perf_counter()), file=sys.stderr)
    gc.callbacks.append(_cb)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 201
hard = len(sys.argv) > 2 and sys.argv[2] == "hard"       # psfm_synth.HARD: every solve rejects steps -> the engine's redo path
import psfm_synth
out = bench_extras.single_sequence_sharded(torch.device("cuda", 0), 0, 1, frames, reps=1 if hard else 3, flows_dist=psfm_synth.HARD if hard else None)
print(json.dumps({k: out.get(k) for k in ("ms_per_sequence", "one_gpu_psfm_connect_ms_per_sequence", "counts_equal_one_gpu",
                                          "solver_counters", "solver_launches", "ms_per_sequence_exchange_form",
                                          "trust_region_iterations", "solves")} |
                 {"frames": frames, "hard": hard, "rounds_ahead": os.environ.get("PSFM_SHARD_ROUNDS_AHEAD", "8")}))
