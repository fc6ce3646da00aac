import sys, time; sys.path.insert(0,'.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
nb,ng,nl=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
pr=acopf.acopf_problem(acopf.synthetic_grid(nb,ng,nl,1),'probe'); print('problem',pr.n,pr.m,pr.nnz)
m=A.Model.from_problem(pr,A.Parameters(algorithm="Line Search",max_iter=10**6))
slp=A.SlpLS(m); t=time.time(); slp.run(max_lp_solves=1); print('time',round(time.time()-t,2))
for r in slp.trace: print('  status',r['status'],{k:v for k,v in r['stats'].items() if k in ('path','ipm_iters','nfact','eqp')})
mdl=pr.model
for nm in ('r_angu','r_angl','r_ref','r_thf','r_tht','r_pb','r_qb','r_pfr','r_qfr','r_pto','r_qto'):
    rr=getattr(mdl,nm); print(nm, rr[0] if len(rr) else None, rr[-1] if len(rr) else None)
