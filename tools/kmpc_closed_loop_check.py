import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from f1tenth_planning_amd.control.kinematic_mpc.kinematic_mpc import KMPCPlanner, mpc_config
g=np.load('/root/repo/tests/golden/tracks.npz'); lev=g['levine']
mpc_line=[lev[:,1],lev[:,2],lev[:,3],lev[:,5]]
def run(seed, host):
    cfgc=mpc_config(); cfgc.SEED=seed
    x=np.array([lev[0,1],lev[0,2],1.0,lev[0,3]])
    pl=KMPCPlanner(waypoints=mpc_line,config=cfgc)
    rng=np.random.default_rng(seed); warm=np.zeros((8,2))
    ds=[]
    for _ in range(30):
        if host:
            ctrl=np.empty((1,8,2,512),np.float32)
            ctrl[0,:,0,:]=rng.normal(0,1.5,(8,512))+warm[:,0:1]; ctrl[0,:,1,:]=rng.normal(0,0.15,(8,512))+warm[:,1:2]
            ctrl[0,:,:,0]=warm; ctrl[0,:,:,1]=0
            out=pl.plan_batch(np.array([x]),controls=ctrl)
            seq=out["best_seq"][0]; warm=np.vstack([seq[1:],seq[-1:]])
            st,sp=float(out["steer"][0]),float(out["speed"][0])
        else:
            st,sp=pl.plan(np.array([x[0],x[1],0.0,x[2],x[3],0.0,0.0]))
        a=(sp-x[2])/cfgc.DTK
        x=pl.predict_motion_kinematic(x,[a],[st])[:,1]
        d=np.hypot(lev[:,1]-x[0],lev[:,2]-x[1]); ds.append(d.min())
    return d.min(), max(ds), int(d.argmin())
for host in (False,True):
    print("host normal" if host else "in-kernel IH-4", [tuple(round(v,3) if isinstance(v,float) else v for v in run(s,host)) for s in range(8)])
