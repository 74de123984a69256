import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import numpy as np, torch, psfm_synth
from point_trajectory import _hip
from point_trajectory.trajectory import run_connect, _result_to_host
ctx=_hip.context()
T,H,W,r=1000,480,640,1
d=psfm_synth.synth_sequence_torch(T,H,W,seed=4,sigma=0.05,n_occluders=2,stride2=False)
res={}
for mode in (1,0):
    ctx.set_chain_mode(mode)
    for it in range(2):
        torch.cuda.synchronize(); t0=time.time()
        info=run_connect(d["flows_f"],d["flows_b"],None,None,3.0,r,return_device=True)
        torch.cuda.synchronize(); t1=time.time()
    print("mode",mode,"chain_mode",info.chain_mode,"%.2f ms"%((t1-t0)*1e3),"traj",info.n_traj,"points",info.n_points,"lanes",info.n_lanes_peak,info.lane_capacity)
    R=_result_to_host(ctx,info); res[mode]=(R.birth.copy(),R.length.copy(),R.xy.copy()); del R
a,b=res[1],res[0]
print("identical", all(np.array_equal(x,y) for x,y in zip(a,b)))
