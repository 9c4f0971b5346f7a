import os, sys, json, numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from tactilesimulation_amd.model.compiler import load_model
from tactilesimulation_amd.model import blob as B
from tactilesimulation_amd.host.batch import BatchSim
from tests.test_gpu_models import _inputs
for name,S in (("tactile_insertion",5),("dclaw_position_control",5)):
    m = load_model(os.path.join(ROOT, "tactilesimulation_amd", "assets", name + ".npz"))
    I=m.I
    print(name, {k:int(I[getattr(B,'TSIM_IH_'+k)]) for k in ['NL','NR','NU','NPAIR','NCPT','NSENSOR','NTAXEL']})
    T=14 if 'ins' in name else 10
    q0,u=_inputs(name,m,64,T)
    sim=BatchSim(m,64,dtype=torch.float64,tape_capacity=0)
    sim.reset(torch.tensor(q0,device='cuda'),None,False)
    ev=[]
    for t in range(T):
        sim.step(torch.tensor(u[:,t],device='cuda'),S); ev.append(sim.last_evals())
    ev=np.array(ev); print(' evals/env-step mean', ev.mean(), 'per step mean', ev.mean(1).round(1), 'max', ev.max(1))
