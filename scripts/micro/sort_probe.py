#!/usr/bin/env python3
"""psfm_sort_records on n random (key, value) pairs: checked against numpy's stable argsort, then run `reps` times (for rocprofv3
--kernel-trace --stats around this script: the per-kernel durations of csrc/psfm_sort.hip).  Usage: sort_probe.py [n] [end_bit] [reps] [check]"""
import ctypes
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import torch
from point_trajectory import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2073277
end_bit = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
check = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(0)
keys = rng.integers(0, 1 << end_bit, n, dtype=np.uint64).astype(np.uint32)
vals = np.arange(n, dtype=np.int32)
ctx = _hip.context(0)
dk = torch.from_numpy(keys.view(np.int32)).cuda()
dv = torch.from_numpy(vals).cuda()
s = _hip.current_stream_ptr()
for r in range(reps):
    k = dk.clone(); v = dv.clone()
    _hip.check(_hip.lib().psfm_sort_records(ctx.handle, _hip.ptr(k), _hip.ptr(v), n, end_bit, s))
torch.cuda.synchronize()
if check:
    order = np.argsort(keys, kind="stable")
    ok = bool(np.array_equal(k.cpu().numpy().view(np.uint32), keys[order]) and np.array_equal(v.cpu().numpy(), vals[order]))
    print("sort_probe n=%d end_bit=%d: %s" % (n, end_bit, "equal to numpy's stable sort" if ok else "DIFFERENT"))
    sys.exit(0 if ok else 1)
