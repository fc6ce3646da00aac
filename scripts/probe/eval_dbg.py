import sys; sys.path.insert(0, '.')
import numpy as np
from tests.test_eval_gpu import _random_function_model, _optimizer_for
fm = _random_function_model(1)
pr = fm.to_problem()
opt = _optimizer_for(pr); opt.eval_setup(fm)
x = np.random.default_rng(51).uniform(-1, 1, pr.n)
f, df, E = opt.eval_functions(x)
Eh = pr.eval_g(x, np.zeros(pr.m))
bad = np.nonzero(E != Eh)[0]
print('bad rows', bad, [(float(E[i]), float(Eh[i]), float(E[i] - Eh[i])) for i in bad])
fs = fm.functions()
for i in bad: print(i, 'const', fs[i].constant, 'aff', fs[i].affine, 'quad', fs[i].quadratic)
