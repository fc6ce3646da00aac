import sys, time; sys.path.insert(0,'.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
name=sys.argv[1]
for ls in [float(a) for a in sys.argv[2:]]:
    for alg in ("Line Search","Trust Region"):
        pr=acopf.acopf_problem(acopf.synthetic_case(name,1,ls),name)
        m=A.Model.from_problem(pr,A.Parameters(algorithm=alg,max_iter=60))
        t=time.time(); s=A.optimize(m); t=time.time()-t
        print(name,'load_scale',ls,alg,'status',m.status,'iter',s.iter,'lp',s.lp_solves,'FR',sum(r['fr'] for r in s.trace),'infeas %.2e'%s.prim_infeas,'obj %.4f'%m.obj_val,'time %.1f'%t, flush=True)
        s.optimizer.close()
