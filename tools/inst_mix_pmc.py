"""Per-kernel dynamic instruction counts from rocprofv3 --pmc passes of the c3
bench (tools/inst_mix_pmc.sh): wave-instructions per launch and per 256 x 256
mode tile, next to the butterfly minimum of that kernel.

Butterfly minimum (packed-f32 instructions per THREAD, fft_radix.h): a radix-16
butterfly = 8 radix-4 (8 packed adds each) + 8 twiddle products (2 packed) = 80;
a 256-point transform = 32 butterflies + 225 inter-stage twiddles over 16 threads
... counted per kernel below as packed instructions per wave and tile:
  forward pass 1      rows: 16 butterflies-equivalents per thread (2 stages x (80 +
                      30 twiddles)) + the column radix-16 (80 + 30) + patch x probe
                      (32): 362 per thread-item, 16 items x 4 waves per tile
  column pass + inverse pass 1 (resident kernel): forward radix-16 (80) + |F|^2
                      (16) + factor (16) + inverse radix-16 + twiddle (110) + the
                      inverse row transform (2 x 110): ~440 per thread and (k1, mode)
  inverse pass 2 + gradients: radix-16 (80) + scale (8) + conj(O) chi (32) +
                      conj(P) chi (32): ~150 per thread and (slice, mode)
"""
import collections
import csv
import glob
import sys

KERNELS = [
    ("fwd_pass1_kernel<256, true>", "tike_fwd_pass1", 362 * 16 * 4),
    ("fwd_grad_ifft2_pass1_resident_kernel", "tike_fwd_grad_ifft2_pass1",
     440 * 16 * 4),
    ("ifft2_pass2_gradients_kernel<256", "tike_ifft2_pass2_gradients",
     150 * 16 * 4),
    ("step_stats_kernel", "tike_lstsq_step_stats", None),
    ("eigen_position_sums1_kernel", "tike_eigen_position_sums1", None),
    ("eigen_pixel_update1_kernel", "tike_eigen_pixel_update1", None),
    ("scatter_patches_kernel", "tike_scatter_patches", None),
]


def collect(d):
    """{kernel substring: {counter: mean per launch over the largest-grid launches}}"""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    # the timed epoch starts at the first solver-only kernel (bench.py's
    # set-up -- `simulate` -- launches the forward kernels too)
    start = min((int(r["Dispatch_Id"]) for r in rows
                 if "psi_precond_kernel" in r["Kernel_Name"]), default=0)
    rows = [r for r in rows if int(r["Dispatch_Id"]) >= start]
    out = {}
    for sub, _, _ in KERNELS:
        mine = [r for r in rows if sub in r["Kernel_Name"]]
        if not mine:
            continue
        g = max(int(r["Grid_Size"]) for r in mine)
        acc = collections.defaultdict(lambda: [0.0, set()])
        for r in mine:
            if int(r["Grid_Size"]) != g:
                continue
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
        out[sub] = {c: v / max(len(ids), 1) for c, (v, ids) in acc.items()}
    return out


def main():
    a, b = collect(sys.argv[1]), collect(sys.argv[2])
    tiles = 1000 * 8  # positions per launch x modes
    print("# Dynamic instruction mix of the c3 minibatch kernels "
          "(1000 positions x 8 modes x 256^2 per launch)\n")
    print("Wave-instructions per launch, in millions (`rocprofv3 --pmc`, two "
          "passes; SQ_INSTS_* count one per wave64 instruction).\n")
    cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
            "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVES"]
    print("| kernel | " + " | ".join(c[3:] for c in cols) + " |")
    print("|---|" + "---|" * len(cols))
    for sub, entry, _ in KERNELS:
        if sub not in a:
            continue
        print(f"| `{entry}` | " + " | ".join(
            f"{a[sub].get(c, 0) / 1e6:.2f}" for c in cols) + " |")
    cols2 = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
             "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
             "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
             "SQ_INSTS_VALU_CVT"]
    print("\nVALU classes (second pass), millions per launch:\n")
    print("| kernel | " + " | ".join(c[14:] for c in cols2) + " |")
    print("|---|" + "---|" * len(cols2))
    for sub, entry, _ in KERNELS:
        if sub not in b:
            continue
        print(f"| `{entry}` | " + " | ".join(
            f"{b[sub].get(c, 0) / 1e6:.2f}" for c in cols2) + " |")
    print("\nVALU wave-instructions per 256 x 256 mode tile against the "
          "butterfly minimum (packed instructions of the transforms and "
          "products the kernel must do, `tools/inst_mix_pmc.py` docstring):\n")
    print("| kernel | VALU per tile | butterfly minimum per tile | ratio |")
    print("|---|---|---|---|")
    for sub, entry, need in KERNELS:
        if sub not in a or need is None:
            continue
        per = a[sub].get("SQ_INSTS_VALU", 0) / tiles
        print(f"| `{entry}` | {per:.0f} | {need} | {per / need:.2f} |")


if __name__ == "__main__":
    main()
