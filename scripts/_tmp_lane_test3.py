import sys,time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, diffsol_amd as H
from diffsol_amd import diffsl
import diffsl_models as D
tol=dict(rtol=1e-6, atol=[1e-6])
nb=262144
cur=np.random.default_rng(12345).uniform(0.6,1.4,(nb,1))
m=diffsl.DiffslModel(D.spm(20))
s=H.Solver(m,cur,nbatch=nb,**tol)
y,tot=s.solve_dense_adaptive([600.0,1800.0,3600.0],want_host=False,group=1)
ts=[]
for _ in range(2):
    t=time.time(); y,tot=s.solve_dense_adaptive([600.0,1800.0,3600.0],want_host=False,group=1); ts.append(time.time()-t)
print("lane solve",round(min(ts),4), flush=True)
