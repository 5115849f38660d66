import numpy as np, sys
sys.path.insert(0,'/root/repo')
import radiativetransfer_amd as rt
from radiativetransfer_amd import synthetic
for nd in (96,192):
    e=rt.DiffuseTransfer()
    n=64
    k,uvb,box=synthetic.uniform_workload(n,8,seed=1,tau_median=0.1)
    e.set_uniform_grid(n,box); e.set_opacity(k)
    ang=np.array([rt.pix2ang_nest(4,i) for i in range(nd)])
    e.transport(ang[:,0].copy(),ang[:,1].copy(),np.full(nd,1/nd),uvb)
    print(nd,"groups",e.counter("brick_groups"),"acc",[e.counter("brick_accumulators_%d"%l) for l in range(3)])
