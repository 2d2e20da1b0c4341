"""Why product and oracle cannot be compared beyond the first epoch on
bench.py's c3 inputs WITH eigen probes ("chaotic, not wrong" must
be a result): the ORACLE against itself, no product code in the runs.

`orthogonalize_eig` (reference probe.py:726-769) takes the probe modes'
eigenvectors from LAPACK, whose phase convention makes the LAST component of
every eigenvector real.  SURVEY 8(d)'s ramp modes give the dominant
eigenvector a last component of 1e-5 and less, so the phase of probe mode 0
-- relative to the eigen probe, which is not rotated with it -- hangs on a
quantity the size of float32 rounding.  Full-size numbers (256^2, 160
positions: +6.5 ... +25 % of the epoch-2 cost for a 3.8e-6 change of the
probe, the measured product-oracle difference after one epoch):
profiles/r06_oracle_sensitivity.txt, tools/oracle_sensitivity.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "tools")):
    if path not in sys.path:
        sys.path.insert(0, path)


def test_oracle_second_epoch_hangs_on_the_eigenvector_phase():
    import bench
    import oracle_sensitivity as T
    from oracle import solvers as osol
    p, ep, ew, data = T.c3_problem(60, det=128, S=8, eigen="init")
    rule = bench.BATCH_RULE
    state, rng, batches = T.first_epoch(osol, p, ep, ew, data, 10, rule)
    v = T.dominant_eigenvector(state["probe"])
    # the component LAPACK makes real is a rounding-sized part of the vector
    assert abs(v[-1].imag) == 0.0 and abs(v[-1]) < 1e-4 < 0.99 < abs(v[0])
    base, _ = T.second_epoch_cost(osol, state, rng, batches, data, rule)
    # an UNSTRUCTURED perturbation of float32 size is not amplified ...
    s2, r2, _ = T.first_epoch(osol, p, ep, ew, data, 10, rule,
                              psi_scale=1 + 1e-6)
    plain, _ = T.second_epoch_cost(osol, s2, r2, batches, data, rule)
    assert abs(plain - base) / base < 2e-4
    # ... a change of the probe of the SAME size along the one direction the
    # phase convention listens to moves the cost a hundred times more
    moved = []
    for theta in (0.0, np.pi / 2):
        c, size = T.second_epoch_cost(osol, state, rng, batches, data, rule,
                                      eps=4e-6, theta=theta)
        assert size < 5e-6
        moved.append(abs(c - base) / base)
    assert max(moved) > 5e-3, moved
    assert max(moved) > 50 * abs(plain - base) / base


def test_oracle_is_stable_without_eigen_probes():
    """The same perturbation without eigen probes: a global phase of a shared
    mode does not change any intensity, the cost does not move."""
    import bench
    import oracle_sensitivity as T
    from oracle import solvers as osol
    p, ep, ew, data = T.c3_problem(40, det=64, S=8, eigen="none")
    rule = bench.BATCH_RULE
    state, rng, batches = T.first_epoch(osol, p, ep, ew, data, 10, rule)
    base, _ = T.second_epoch_cost(osol, state, rng, batches, data, rule)
    c, _ = T.second_epoch_cost(osol, state, rng, batches, data, rule,
                               eps=4e-6, theta=np.pi / 2)
    assert abs(c - base) / base < 2e-4
