import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
import activesetmethods_amd as A
from collections import Counter
pr=A.problems.synthetic_dense_nlp(1000,500)
for alg in ("Trust Region","Line Search"):
    m=A.Model.from_problem(pr,A.Parameters(algorithm=alg,max_iter=int(sys.argv[1])))
    t=time.time(); s=A.optimize(m); t=time.time()-t
    print(alg,'ret',m.status,'iter',s.iter,'lp',s.lp_solves,'obj',m.obj_val,'infeas',s.prim_infeas,'time',round(t,2),'lp_time',round(s.lp_time,2))
    print(' paths',Counter(r['stats']['path'] for r in s.trace),'ipm iters',sum(r['stats']['ipm_iters'] for r in s.trace),'nfact',sum(r['stats']['nfact'] for r in s.trace))
    ks=s.optimizer.kernel_stats()
    for k,v in ks.items(): print('  ',k,{a:(round(b,3) if a=='ms' else b) for a,b in v.items()})
