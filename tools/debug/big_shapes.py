import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, time
import tike_amd.ptycho as tp
from test_solvers_gpu import _minibatch_vs_oracle
from tike_amd.ptycho.solvers._plan import GradientPlan
real = GradientPlan.gradients
routes = []
def spy(self, c, k):
    routes.append(self.route); return real(self, c, k)
GradientPlan.gradients = spy
for det, pw, S, N in ((1536, 1536, 1, 2), (2048, 2048, 1, 2), (2048, 1500, 2, 2), (1280, 1280, 1, 2), (1792, 1792, 1, 2)):
    t = time.time(); del routes[:]
    try:
        _minibatch_vs_oracle(tp, det, S, N, False, pw=pw)
        print("ok ", det, pw, S, N, sorted(set(routes)), f"{time.time() - t:.1f} s", flush=True)
    except Exception as e:
        print("ERR", det, pw, S, N, sorted(set(routes)), type(e).__name__, str(e)[:300], flush=True)
