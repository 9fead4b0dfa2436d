import numpy as np, torch, sys
sys.path.insert(0,'.')
import retto_amd
from retto_amd import synth
from oracle import nets_torch as N
s=retto_amd.RettoSession(retto_amd.synthetic_session_config(0))
d,c,r,dic=synth.synth_models(0)
wd=N.read_blob(d); wd64={k:v.double() for k,v in wd.items()}
for (n,h,w) in [(1,64,96),(2,160,128),(1,320,320),(1,960,960)]:
    x=np.random.default_rng(h+w).uniform(-1,1,(n,3,h,w)).astype(np.float32)
    got=s.worker.det(x)
    ref32=N.det_forward(wd,torch.from_numpy(x)).numpy()
    ref64=N.det_forward(wd64,torch.from_numpy(x).double()).numpy()
    e_hip=np.abs(got-ref64); e_t=np.abs(ref32-ref64)
    print((n,h,w),'hip vs f64 max %.2e mean %.2e | torch32 vs f64 max %.2e mean %.2e | hip vs torch32 max %.2e'%(e_hip.max(),e_hip.mean(),e_t.max(),e_t.mean(),np.abs(got-ref32).max()))
