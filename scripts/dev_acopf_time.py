import sys, time; sys.path.insert(0,'.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
from collections import Counter
name=sys.argv[1]; nlp=int(sys.argv[2]); alg=sys.argv[3] if len(sys.argv)>3 else "Line Search"
t=time.time(); pr=acopf.acopf_problem(acopf.synthetic_case(name,1),name); print('problem',pr.n,pr.m,pr.nnz,'built in',round(time.time()-t,2))
m=A.Model.from_problem(pr,A.Parameters(algorithm=alg,max_iter=10**6))
slp=A.SlpLS(m) if alg=="Line Search" else A.SlpTR(m)
t=time.time(); slp.run(max_lp_solves=nlp); t=time.time()-t
print(alg,'lp solves',slp.lp_solves,'time',round(t,2),'lp_time',round(slp.lp_time,2),'infeas',slp.prim_infeas)
for r in slp.trace: print('  status',r['status'],'fr',r['fr'],{k:(round(v,3) if isinstance(v,float) else v) for k,v in r['stats'].items()})
ks=slp.optimizer.kernel_stats()
for k,v in ks.items():
    rate = v['flops']/(v['ms']*1e-3)/1e12 if v['ms']>0 else 0
    bw = v['bytes']/(v['ms']*1e-3)/1e9 if v['ms']>0 else 0
    print('  %-9s ms %10.2f calls %6d  TFLOP/s %7.2f  GB/s %8.1f'%(k,v['ms'],v['calls'],rate,bw))
