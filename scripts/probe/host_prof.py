"""cProfile of the host side of SLP steps (development probe; GPU box).  usage: host_prof.py c3|c5 STEPS"""
import sys, cProfile, pstats; sys.path.insert(0, '.')
import activesetmethods_amd as A
from activesetmethods_amd import acopf
name = {"c3": "case118", "c5": "case300"}[sys.argv[1]]; steps = int(sys.argv[2])
fm = acopf.function_model(acopf.synthetic_case(name, 1, 0.5 if sys.argv[1] == "c5" else 1.0))
pr = fm.to_problem(name)
m = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=10 ** 9, device_eval=True))
slp = A.SlpLS(m)
slp.run(max_lp_solves=3)
prof = cProfile.Profile()
prof.enable()
slp.run(max_lp_solves=3 + steps, resume=True)
prof.disable()
st = pstats.Stats(prof); st.sort_stats("tottime").print_stats(18)
