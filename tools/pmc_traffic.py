#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/rNN_pmc_traffic_<workload>.json.

python tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> <positions_per_launch> <out.json>

Per kernel, only the launches with the largest grid are kept (the timed
epoch's full chunks; set-up launches are smaller) and averaged.  HBM bytes =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B
request; WRITE_SIZE is exact) as MI355X_MICROARCH.md prescribes.
"""
import collections
import csv
import glob
import json
import sys

# kernel-name prefix -> C-ABI entry whose launch it is
KERNELS = {
    # (the poisson sweeps inside the resident gradient kernel: before its plain form)
    "void fwd_grad_ifft2_pass1_resident_kernel<4, 1, float, 1,": "tike_poisson_steps_grad_ifft2_pass1:sweep1",
    "void fwd_grad_ifft2_pass1_resident_kernel<4, 1, float, 2,": "tike_poisson_steps_grad_ifft2_pass1:sweep2",
    "void fwd_grad_ifft2_pass1_resident_kernel<3, 1, float, 1,": "tike_poisson_steps_grad_ifft2_pass1:sweep1",
    "void fwd_grad_ifft2_pass1_resident_kernel<3, 1, float, 2,": "tike_poisson_steps_grad_ifft2_pass1:sweep2",
    "void ptycho_fwd_pos_kernel<256, false>": "tike_ptycho_fwd_gradient_scale",
    "void ptycho_fwd_pos_kernel<256, true>": "tike_ptycho_fwd_intensity",
    "void ptycho_fwd_pos_kernel<512": "tike_ptycho_fwd_intensity",
    "void fwd_pass1_kernel": "tike_fwd_pass1",
    "void pfa_fwd_gather_kernel": "tike_pfa_fwd_gather",
    "void pfa_combine_gradient_kernel": "tike_pfa_combine_gradient",
    "void pfa_inv_products_kernel": "tike_pfa_inv_products",
    "void fft2_v2_kernel": "tike_pfa_fft2",
    "gen_fwd_rows_kernel": "tike_gen_fwd_rows",
    "void gen_cols_resident_kernel": "tike_gen_cols_gradient",
    "void gen_cols_gradient_kernel": "tike_gen_cols_gradient",
    "gen_inv_rows_gradients_kernel": "tike_gen_inv_rows_gradients",
    "void fwd_colpass_inplace_kernel": "tike_ptycho_fwd",
    "void fwd_gradient_scale_kernel": "tike_fwd_gradient_scale",
    "void fwd_grad_ifft2_pass1_kernel": "tike_fwd_grad_ifft2_pass1",
    "void fwd_grad_ifft2_pass1_single_kernel": "tike_fwd_grad_ifft2_pass1",
    "void fwd128_lds_kernel": "tike_ptycho_fwd",
    "void farplane_gradient_kernel": "tike_farplane_gradient",
    "eigen_pixel_update1_kernel": "tike_eigen_pixel_update1",
    "eigen_position_sums1_kernel": "tike_eigen_position_sums1",
    "eigen_position_sums1_pair_kernel": "tike_eigen_position_sums1",
    "ls_trial_kernel": "tike_cgrad_line_search:trial",
    "ls_decide_kernel": "tike_cgrad_line_search:decide",
    "void ls_ksteps_colpass_kernel": "tike_cgrad_line_search_linear:costs",
    "void ls_ksteps_farplane_kernel": "tike_cgrad_line_search_linear:costs",
    "ls_pick_kernel": "tike_cgrad_line_search_linear:pick",
    "ls_apply_kernel": "tike_cgrad_line_search_linear:apply",
    "void fwd_grad_ifft2_pass1_resident_kernel": "tike_fwd_grad_ifft2_pass1",
    "void poisson_sweep2_grad_ifft2_pass1_kernel": "tike_poisson_steps_grad_ifft2_pass1:sweep2",
    "void poisson_colpass_kernel": "tike_poisson_steps:sweep",
    "void fwd_grad_ifft2_pass1_512_kernel": "tike_fwd_grad_ifft2_pass1",
    "void plain_pass1_kernel": "tike_ptycho_adj:pass1",
    "void ifft2_pass2_adjoint_kernel": "tike_ptycho_adj:pass2",
    "adj_interleave_kernel": "tike_ptycho_adj:interleave",
    "void grad_ifft2_pass1_512_kernel": "tike_grad_ifft2_pass1",
    "void grad_ifft2_crop_kernel<256, 1, false>": "tike_grad_ifft2_pass1",
    "void grad_ifft2_crop_kernel<256, 2, false>": "tike_grad_ifft2_pass1",
    "void grad_ifft2_crop_kernel<256, 1, true>": "tike_grad_ifft2_crop",
    "void grad_ifft2_crop_kernel<256, 2, true>": "tike_grad_ifft2_crop",
    "void ifft2_pass2_gradients_kernel": "tike_ifft2_pass2_gradients",
    "void ifft2_crop_v2_kernel<128, 1, false>": "tike_ifft2_pass1_scaled",
    "void ifft2_crop_v2_kernel<256, 1, false>": "tike_ifft2_pass1_scaled",
    "void ifft2_crop_v2_kernel<512, 1, false>": "tike_ifft2_pass1_scaled",
    "void ifft2_crop_v2_kernel": "tike_ifft2_crop_scaled",
    "void probe_grad_kernel<true": "tike_lstsq_gradients",
    "void step_stats_kernel": "tike_lstsq_step_stats",
    "void step_stats_pair_kernel": "tike_lstsq_step_stats",
    "scatter_patches_kernel": "tike_scatter_patches",
    "void gradient_scale_kernel": "tike_gradient_scale",
    "psi_precond_kernel": "tike_psi_preconditioner",
    "eigen_position_sums_kernel": "tike_eigen_position_sums",
    "eigen_pixel_update_kernel": "tike_eigen_pixel_update",
    "varying_probe_kernel": "tike_varying_probe",
    "position_sums_kernel": "tike_position_sums",
}


# kernels only the solver launches: the first of them marks the start of the
# timed step (bench.py's set-up -- `simulate` -- runs the forward operator,
# whose kernels the solver uses too)
SOLVER_ONLY = ("psi_precond_kernel", "void fwd_gradient_scale_kernel",
               "void fwd_grad_ifft2_pass1_kernel",
               "void fwd_grad_ifft2_pass1_resident_kernel",
               "void poisson_colpass_kernel",
               "void step_stats_kernel", "void step_stats_pair_kernel",
               "void probe_grad_kernel",
               "void gradient_scale_kernel", "void farplane_gradient_kernel",
               "void fwd_grad_ifft2_pass1_single_kernel", "ls_trial_kernel",
               "void fwd_grad_ifft2_pass1_512_kernel",
               "void ifft2_pass2_adjoint_kernel")


def collect(d, counter, cgrad=False):
    rows = collections.defaultdict(list)  # entry -> [(grid, value)]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        table = list(csv.DictReader(open(f)))
        starts = [int(r["Dispatch_Id"]) for r in table
                  if r["Kernel_Name"].startswith(SOLVER_ONLY)]
        first = min(starts) if starts else 0
        # cgrad: the forward kernels run both in the gradient pass (followed by
        # the gradient kernels) and in cost-only line-search trials (followed by
        # the cost kernel and the decision): told apart by what comes next
        order = sorted({(int(r["Dispatch_Id"]), r["Kernel_Name"])
                        for r in table if "fillBuffer" not in r["Kernel_Name"]})
        nxt = {order[i][0]: order[i + 1][1] for i in range(len(order) - 1)}
        for r in table:
            if r["Counter_Name"] != counter or int(r["Dispatch_Id"]) < first:
                continue
            for prefix, entry in KERNELS.items():
                if r["Kernel_Name"].startswith(prefix):  # first match wins
                    if cgrad and entry == "tike_fwd_pass1" and nxt.get(
                            int(r["Dispatch_Id"]), "").startswith(
                                ("void fwd_gradient_scale_kernel",
                                 "void ls_ksteps_colpass_kernel")):
                        entry += ":cost_only"  # (of a trial / of the direction)
                    if cgrad and entry == "tike_ptycho_fwd" and nxt.get(
                            int(r["Dispatch_Id"]), "").startswith(
                                "void ls_ksteps_farplane_kernel"):
                        entry += ":direction"
                    if cgrad and entry == "tike_fwd_gradient_scale":
                        entry += ":cost_only"
                    rows[entry].append((int(r["Grid_Size"]),
                                        float(r["Counter_Value"])))
                    break
    out = {}
    for entry, v in rows.items():
        g = max(x[0] for x in v)
        vals = [x[1] for x in v if x[0] == g]
        out[entry] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    fetch_dir, write_dir, workload, n, out = sys.argv[1:6]
    cgrad = workload in ("c1", "c2")
    fetch = collect(fetch_dir, "FETCH_SIZE", cgrad)
    write = collect(write_dir, "WRITE_SIZE", cgrad)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tike_amd._lib import ABI_VERSION, build_id
    doc = {
        "workload": workload,
        "positions_per_launch": int(n),
        # the kernels these counters belong to (bench.measured_traffic quotes
        # the file only while the loaded library is this build)
        "build_id": build_id(),
        "abi_version": ABI_VERSION,
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on "
                "`python3 bench.py --no-cpu-baseline --steps 1 --warmup 0`, "
                "averaged over the launches with the largest grid (the timed "
                "epoch); hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: "
                "FETCH_SIZE counts 64 B per 128-B request)",
        "kernels": {},
    }
    for entry in sorted(set(fetch) & set(write)):
        f, nf = fetch[entry]
        w, _ = write[entry]
        doc["kernels"][entry] = {
            "fetch_kib_per_launch": f,
            "write_kib_per_launch": w,
            "launches": nf,
            "hbm_bytes_per_launch": (2 * f + w) * 1024,
        }
    # one timed step = launches-per-step x bytes per launch, summed over the
    # kernels seen (bench.py runs --steps 1 --warmup 0 under the counters, so
    # the launch count of the largest-grid launches is the count per step)
    # bench.py's set-up (`simulate`: the far-plane-storing forward operator)
    # is not part of a step
    setup = {"c3": ("tike_ptycho_fwd_intensity",),
             "c2": ("tike_ptycho_fwd_intensity",),
             "c5": ("tike_ptycho_fwd_intensity",)}.get(workload, ())
    if workload == "c2":
        # the gradient of a chunk is ONE C-ABI call made of four launches
        parts = ("tike_fwd_pass1", "tike_fwd_grad_ifft2_pass1",
                 "tike_ifft2_pass2_gradients", "tike_scatter_patches")
        if all(k in doc["kernels"] for k in parts):
            doc["composite"] = {"tike_lstsq_chunk_gradients": {
                "parts": list(parts),
                "hbm_bytes_per_launch": sum(
                    doc["kernels"][k]["hbm_bytes_per_launch"] for k in parts)}}
    if workload == "c3poisson":
        # one call = first sweep + (second sweep + gradient pass)
        parts = ("tike_poisson_steps_grad_ifft2_pass1:sweep1",
                 "tike_poisson_steps_grad_ifft2_pass1:sweep2")
        if all(k in doc["kernels"] for k in parts):
            doc["composite"] = {"tike_poisson_steps_grad_ifft2_pass1": {
                "parts": list(parts),
                "hbm_bytes_per_launch": sum(
                    doc["kernels"][k]["hbm_bytes_per_launch"] for k in parts)}}
    if workload.startswith("fwd"):
        # one Ptycho.fwd call = forward pass 1 + the column pass in place per
        # sub-batch (256^2 / 512^2), or the one whole-tile launch (128^2)
        parts = [k for k in ("tike_fwd_pass1", "tike_ptycho_fwd")
                 if k in doc["kernels"]]
        # (`--steps 1 --warmup 0` under the counters: one call)
        doc["composite"] = {"tike_ptycho_fwd": {
            "parts": parts,
            "hbm_bytes_per_launch": sum(
                doc["kernels"][k]["hbm_bytes_per_launch"] *
                doc["kernels"][k]["launches"] for k in parts)}}
    if workload.startswith("adj"):
        # one Ptycho.adj call = the launches of all its sub-batches: inverse
        # pass 1, pass 2 with both products, the grouped scatter, one
        # interleave (the first pass-1 launch precedes the marker kernel:
        # counted by its launches-per-call below)
        parts = [k for k in doc["kernels"]
                 if k.startswith("tike_ptycho_adj:") or k == "tike_scatter_patches"]
        calls = max(1, doc["kernels"].get("tike_ptycho_adj:interleave",
                                          {}).get("launches", 1))
        # (every sub-batch has one pass 1, one pass 2 and one scatter)
        per_call = doc["kernels"].get("tike_ptycho_adj:pass2", {}).get("launches", 0)
        count = lambda k: (per_call if k != "tike_ptycho_adj:interleave"
                           else doc["kernels"][k]["launches"])
        doc["composite"] = {"tike_ptycho_adj": {
            "parts": parts,
            "sub_batches_per_call": per_call / calls,
            "hbm_bytes_per_launch": sum(
                doc["kernels"][k]["hbm_bytes_per_launch"] * count(k)
                for k in parts) / calls}}
    doc["setup_kernels_excluded_from_step"] = list(setup)
    doc["hbm_bytes_per_step"] = sum(
        k["hbm_bytes_per_launch"] * k["launches"]
        for name, k in doc["kernels"].items() if name not in setup)
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in doc["kernels"].items():
        print(f"{k:34s} read {2*v['fetch_kib_per_launch']/1048576:7.2f} GiB  "
              f"write {v['write_kib_per_launch']/1048576:7.2f} GiB")


if __name__ == "__main__":
    main()
