#!/usr/bin/env python3
"""Stand-alone flow_check at the headline shape: GB/s of the kernel PSFM_FC_KERNEL selects (1: four strided pixels per
thread, 8-byte loads; 2: two adjacent pixels per lane, 16-byte loads), mask hash for cross-checking the two."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    sys.path.insert(0, p)
import torch
import psfm_synth
from point_trajectory.utils import flow_check_device
H, W, T = 1080, 1920, 101
d = psfm_synth.synth_sequence_torch(T, H, W, seed=0, sigma=0.05, n_occluders=2, stride2=False, device="cuda")
for _ in range(3):
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    _, occ = flow_check_device(d["flows_f"], d["flows_b"], 1.0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
bytes_ = 17.0 * H * W * (T - 1)
print(json.dumps({"kernel": os.environ.get("PSFM_FC_KERNEL", "2"), "ms_per_100_pairs": ms, "GBs": bytes_ / (ms * 1e-3) / 1e9,
                  "frac_of_8TBs": bytes_ / (ms * 1e-3) / 8e12, "occluded": int(occ.sum().item()),
                  "mask_sha": hashlib.sha256(occ.cpu().numpy().tobytes()).hexdigest()[:16]}))
