"""Epoch rate of the off-grid workloads (bench.py c3pad / c3m12 / c384: probe
window < detector, 12 modes, a 384^2 detector) beside c3's, as far-plane bytes
per second (T = 8 S det^2 per position) -- VERDICT r5 item 1(c): "the cliff
is a number".

    gpurun -- python tools/offgrid_legs.py [positions=2000] [workload ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

if os.environ.get("OFFGRID_GENERAL") == "0":  # the unfused round-1 kernels
    from tike_amd.ptycho.solvers import lstsq as _L
    _L.GENERAL_FUSED = False
import re
for name in sys.argv[2:]:  # gDETxS: any detector size / mode count, no eigen probe
    # (c3gDETxS: the same with the eigen probe of the c3 workloads)
    m = re.fullmatch(r"(?:c3)?g(\d+)x(\d+)", name)
    if m:
        bench.EPOCH_DEFAULTS[name] = (int(m.group(1)), int(m.group(2)), 10000, 10)
if os.environ.get("OFFGRID_GROUPS") == "0":  # > 8 modes: the stored far plane
    from tike_amd.ptycho.solvers import lstsq as _L3
    _L3.MODE_GROUPS = False
if os.environ.get("OFFGRID_PFA") == "0":  # the LDS line engine instead
    from tike_amd.ptycho.solvers import lstsq as _L2
    _L2.PFA_ROUTE = False
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for w in sys.argv[2:] or ["c3", "c3pad", "c3m12", "c384"]:
    leg = bench.epoch_leg(w, tp, A, torch, positions=N, epochs=3)
    T = 8 * leg["modes"] * leg["detector"]**2
    leg["farplane_GBs"] = T * leg["value"] / 1e9
    print(json.dumps(leg), flush=True)
    print(f"{w:8s} {leg['value'] / 1e3:8.1f} k patterns/s  "
          f"{leg['farplane_GBs']:7.1f} GB/s of far plane", flush=True)
