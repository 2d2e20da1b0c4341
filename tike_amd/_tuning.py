"""Every A/B lever of the package that is read from the environment, in one
place (seventeen `TIKE_*` switches were read inside product
modules).  The modules import the VALUES below once, at import time, into the
module attributes tests patch (`lstsq.POISSON_STEPS_IN_PASS2`,
`rpie.STEP_BACK_IN_FREQUENCY`, ...); nothing under `solvers/` or `operators/`
touches `os.environ`.  Defaults are what the driver's bench runs; the
non-default values are kept for same-box A/B runs (`tools/ab_env.sh`) and for
the tests that compare two routes.

  TIKE_ONE_LAUNCH_512=0     512^2: column pass + factor and inverse pass 1 as
                            two launches (round 4) instead of one
  TIKE_POISSON_LINEAR=0     poisson, 256^2: step lengths applied before the
                            inverse instead of by pass 2
  TIKE_CHUNK_POSITIONS=n    kernel chunks of n positions (several per minibatch)
  TIKE_EIGEN_SUMS_GATHER=0  eigen position sums on the stored patches
  TIKE_STATS_GATHER=1       step statistics gather O_n from the object
  TIKE_PRECOND_CHUNK=n      positions per chunk of the multislice preconditioner
  TIKE_MS_SLICE_STEP=0 / TIKE_MS_FIRST_STORED=0 / TIKE_MS_STEP_BACK=0
                            multislice rpie: the unfused slice step / the
                            gathered first slice / the step back as written
  TIKE_CGRAD_GRAPHS=1       cgrad: replay captured HIP graphs (slower on ROCm 7.2)
  TIKE_FWD_SUB_MIB=n        Ptycho.fwd / adj: far-plane MiB per sub-batch
"""
import os


def _flag(name, default):
    return os.environ.get(name, "1" if default else "0") == "1"


def _int(name, default=None):
    v = os.environ.get(name)
    return int(v) if v else default


one_launch_512 = _flag("TIKE_ONE_LAUNCH_512", True)
poisson_steps_in_pass2 = _flag("TIKE_POISSON_LINEAR", True)
chunk_positions = _int("TIKE_CHUNK_POSITIONS")
eigen_sums_gather = _flag("TIKE_EIGEN_SUMS_GATHER", True)
stats_gather = _flag("TIKE_STATS_GATHER", False)
precond_chunk = _int("TIKE_PRECOND_CHUNK", 512)
multislice_slice_step = _flag("TIKE_MS_SLICE_STEP", True)
multislice_first_stored = _flag("TIKE_MS_FIRST_STORED", True)
multislice_step_back = _flag("TIKE_MS_STEP_BACK", True)
cgrad_graphs = _flag("TIKE_CGRAD_GRAPHS", False)
fwd_sub_mib = (float(os.environ["TIKE_FWD_SUB_MIB"])
               if os.environ.get("TIKE_FWD_SUB_MIB") else None)
