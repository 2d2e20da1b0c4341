#!/usr/bin/env python3
"""Golden vectors for the probe helpers (build container only):

    python tests/golden/gen/make_probe_fixtures.py

* `probe_constraints.npz`: the REFERENCE's constrain_center_peak,
  apply_median_filter_abs_probe and constrain_probe_sparsity run (under the
  NumPy-backed CuPy stand-in, whose cupyx.scipy.ndimage forwards to SciPy) on
  seeded probes.  CuPy's median_filter converts its window size to integers
  (cupyx/scipy/ndimage/_filters_core.py `_fix_sequence_arg(size, ndim, 'size',
  int)`); SciPy refuses floats, so the window is handed over as integers.
* `fresnel_probes.npz`: the REFERENCE's single_probe / MW_probe
  (tike.ptycho.fresnel) for custom and preset zone plates, forward and
  backward propagation, a Gaussian line and a supplied spectrum.
* `ref_hermite.npz`: tests/ptycho/hermite.mat (the data file of the
  reference's own test_hermite_modes) re-encoded.
"""
import os
import sys

sys.dont_write_bytecode = True  # never write into /root/reference

import numpy as np
import scipy.io

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "cupy_shim"))
sys.path.insert(0, os.path.join(REF, "src"))
os.environ.setdefault("TIKE_REF_EMU_LIB", "")

import cupy as cp  # noqa: E402,F401  (the shim)
import cupyx.scipy.ndimage as _cnd  # noqa: E402
import scipy.ndimage as _snd  # noqa: E402
import tike.ptycho.probe as ref  # noqa: E402

# cupy.unravel_index names its second argument `dims` (NumPy: `shape`)
cp.unravel_index = lambda indices, dims, order="C": np.unravel_index(
    indices, dims, order=order)
# CuPy semantics for the window size (integers), see the module docstring
_cnd.median_filter = lambda input, size, **kw: _snd.median_filter(
    input, size=tuple(int(v) for v in size), **kw)

rng = np.random.default_rng(77)
out = {}


def blob(h, w, cy, cx, s, modes):
    y, x = np.mgrid[:h, :w]
    amp = np.exp(-((y - cy)**2 + (x - cx)**2) / (2 * s * s))
    return np.stack([
        amp * (0.2 + rng.random((h, w))) * np.exp(2j * np.pi * rng.random((h, w)))
        / (m + 1) for m in range(modes)
    ])[None, None].astype(np.complex64)


# centre constraint: peaks off-centre in different directions (+ one centred)
for name, (h, w, cy, cx) in dict(a=(32, 32, 9, 22), b=(31, 33, 20, 10),
                                 c=(24, 24, 12, 12), d=(16, 16, 8, 3)).items():
    p = blob(h, w, cy, cx, 3.0, 3)
    out[f"center_in_{name}"] = p
    out[f"center_out_{name}"] = np.asarray(ref.constrain_center_peak(p.copy()))
# median filter of the amplitude: odd, even and mixed windows
p = blob(20, 23, 10, 11, 5.0, 2)
p[0, 0, 0, 4, 5] *= 40  # hot spots
p[0, 0, 1, 15, 2] *= 25
out["median_in"] = p
for a, b in ((3, 3), (2, 4), (1, 5), (1, 1)):
    out[f"median_out_{a}_{b}"] = np.asarray(
        ref.apply_median_filter_abs_probe(p.copy(), med_filt_px=(a, b)))
# sparsity
p = blob(32, 32, 14, 18, 4.0, 2)
out["sparse_in"] = p
for f in (0.25, 0.6):
    out[f"sparse_out_{int(f * 100)}"] = np.asarray(
        ref.constrain_probe_sparsity(p.copy(), f))
# scanning-transmission image and Gaussian derivatives (host helpers)
import tike.ptycho.object as ref_object  # noqa: E402
import tike.ptycho.position as ref_position  # noqa: E402
scan = (rng.random((40, 2)) * (9, 13) + 2).astype(np.float32)
data = rng.random((40, 8, 8)).astype(np.float32)
out["absorb_scan"], out["absorb_data"] = scan, data
for method in ("cubic", "linear", "nearest"):
    out[f"absorb_{method}"] = ref_object.get_absorbtion_image(
        data, scan, rescale=0.8, method=method)
x = rng.normal(size=(3, 11, 9)).astype(np.float32)
out["gradient_in"] = x
gy, gx = ref_position.gaussian_gradient(cp.asarray(x))
out["gradient_y"], out["gradient_x"] = np.asarray(gy), np.asarray(gx)
path = os.path.join(OUT, "probe_constraints.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) / 1e6, "MB")

# Fresnel zone-plate probes (tike.ptycho.fresnel): pure NumPy in the reference
import tike.ptycho.fresnel as ref_fresnel  # noqa: E402
lam = 1.24e-9 / 10
dx = lam * 2 / 64 / 75e-6
plate = dict(radius=150e-6 / 2, outmost=50e-9, beamstop=60e-6)
spec = np.stack([lam * (1 + 0.002 * np.arange(-4, 5)),
                 np.exp(-np.arange(-4, 5)**2 / 8.0)], 1)
fres = dict(
    params=np.array([lam, dx]), spectrum=spec,
    single=ref_fresnel.single_probe(64, lam, dx, 800e-6, zone_plate_params=plate),
    single_velo_back=ref_fresnel.single_probe(48, lam, dx, -300e-6,
                                              zone_plate_params="velo"),
    mw_2idd=ref_fresnel.MW_probe(64, lam, dx, 800e-6, zone_plate_params="2idd",
                                 energy=5, bandwidth=0.01),
    mw_lamni_spectrum=ref_fresnel.MW_probe(64, lam, dx, 800e-6,
                                           zone_plate_params="lamni", energy=3,
                                           spectrum=spec.copy()))
path = os.path.join(OUT, "fresnel_probes.npz")
np.savez_compressed(path, **fres)
print(path, os.path.getsize(path) / 1e6, "MB")

m = scipy.io.loadmat(f"{REF}/tests/ptycho/hermite.mat")
path = os.path.join(OUT, "ref_hermite.npz")
np.savez_compressed(path, probes=m["probes"], result=m["result"])
print(path, os.path.getsize(path) / 1e6, "MB", m["probes"].shape,
      m["result"].shape, m["probes"].dtype)
