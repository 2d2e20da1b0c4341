#!/usr/bin/env python3
"""Golden vectors for the data readers: the REFERENCE's
``tike.ptycho.io.read_aps_velociprobe`` / ``read_aps_lynx`` run on synthetic
files (build container only; needs /root/reference).

    python tests/golden/gen/make_io_fixtures.py

h5py is absent from the image, so the reference is handed ``tests/fake_h5.py``
as its ``h5py`` module (the readers only open a file, index datasets and read
one attribute); position files are real text files.  Stored: the synthetic
inputs and the reference's outputs -- data only.
"""
import os
import sys

sys.dont_write_bytecode = True  # never write into /root/reference
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
sys.path.insert(0, os.path.dirname(OUT))  # tests/
sys.path.insert(0, os.path.join(HERE, "cupy_shim"))
sys.path.insert(0, "/root/reference/src")

import fake_h5  # noqa: E402

TREES = {}
h5py = types.ModuleType("h5py")
h5py.File = lambda path, mode="r": fake_h5.File(TREES[path])
sys.modules["h5py"] = h5py

import tike.ptycho.io as ref_io  # noqa: E402

rng = np.random.default_rng(2024)
tmp = tempfile.mkdtemp(prefix="tike_io_")
out = {}

# ---- Velociprobe: two linked files (+ one dangling), 8-column CSV with
# several interferometer samples per trigger, split over two files
H, W = 70, 90
frames = [rng.integers(0, 4000, (n, H, W)).astype(np.uint16) for n in (5, 4)]
meta = dict(photon_energy=8800.0, beam_center=(47, 33), distance=1.92,
            pixel_size=75e-6, chi=12.5)
TREES["velo.h5"] = fake_h5.velociprobe_tree(frames, **meta)
rows = []
for trig in range(10):  # one position more than there are frames
    for _ in range(int(rng.integers(1, 5))):
        r = rng.integers(-2_000_000, 2_000_000, 8)
        r[7] = trig
        rows.append(r)
rows = np.array(rows, dtype=np.int64)
csv = [os.path.join(tmp, "pos0.csv"), os.path.join(tmp, "pos1.csv")]
cut = int(np.searchsorted(rows[:, 7], 6))
np.savetxt(csv[0], rows[:cut], fmt="%d", delimiter=",")
np.savetxt(csv[1], rows[cut:], fmt="%d", delimiter=",")
out.update(velo_frames0=frames[0], velo_frames1=frames[1], velo_rows=rows,
           velo_cut=cut,
           velo_meta=np.array([meta["photon_energy"], *meta["beam_center"],
                               meta["distance"], meta["pixel_size"],
                               meta["chi"]]))
import warnings  # noqa: E402
for binned in (1, 2):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d, s = ref_io.read_aps_velociprobe("velo.h5", csv, binned_pix=binned)
    out[f"velo_data_b{binned}"] = d
    out[f"velo_scan_b{binned}"] = s
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    d, s = ref_io.read_aps_velociprobe("velo.h5", csv[0], xy_columns=(6, 4),
                                       max_crop=16)
out["velo_data_crop16"], out["velo_scan_crop16"] = d, s

# ---- LYNX: one dataset with detector gaps, .dat with two header rows
H, W = 80, 100
eiger = rng.integers(0, 3000, (6, H, W)).astype(np.uint16)
eiger[:, :, 50:53] = 2**12 - 1
eiger[:, 38:40, :] = 2**12 - 1
TREES["lynx.h5"] = fake_h5.lynx_tree(eiger, 75e-6)
dat = os.path.join(tmp, "scan.dat")
table = np.round(rng.normal(0, 3, (6, 8)), 4)
table[:, 0] = np.arange(6)
with open(dat, "w") as f:
    f.write("# LYNX scan\n# n a b x c d y e\n")
    np.savetxt(f, table, fmt="%.4f", delimiter=" ")
lynx = dict(photon_energy=9000.0, beam_center_x=52, beam_center_y=41,
            detector_dist=2.3)
out.update(lynx_frames=eiger, lynx_table=table,
           lynx_meta=np.array([lynx["photon_energy"], lynx["beam_center_x"],
                               lynx["beam_center_y"], lynx["detector_dist"],
                               75e-6]))
for binned in (1, 4):
    d, s = ref_io.read_aps_lynx("lynx.h5", dat, binned_pix=binned, **lynx)
    out[f"lynx_data_b{binned}"] = d
    out[f"lynx_scan_b{binned}"] = s
out["units"] = ref_io.position_units_to_pixels(
    np.array([[1e-6, -2e-6], [0.5e-6, 3e-6]]), 1.92, 256, 75e-6, 8800.0)

path = os.path.join(OUT, "io_readers.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path) / 1e6, "MB")
for k, v in out.items():
    print(k, np.asarray(v).shape, np.asarray(v).dtype)
