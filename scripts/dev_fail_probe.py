import sys, time; sys.path.insert(0,'.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
name=sys.argv[1]; ls=float(sys.argv[2]); alg=sys.argv[3]
pr=acopf.acopf_problem(acopf.synthetic_case(name,1,ls),name)
m=A.Model.from_problem(pr,A.Parameters(algorithm=alg,max_iter=60))
s=A.optimize(m)
print('status',m.status,'iter',s.iter,'lp',s.lp_solves)
for k,r in enumerate(s.trace[-4:]):
    print(len(s.trace)-4+k, 'fr', r['fr'], 'status', r['status'], r['stats'])
