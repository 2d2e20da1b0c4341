"""Static instruction mix of the gfx950 kernels: classes of instructions per
kernel (or per labelled loop of a kernel) from the device assembly.

    hipcc -O3 --offload-arch=gfx950 -S --offload-device-only x.hip -o x.s
    python tools/inst_mix.py x.s [substring-of-kernel-name ...] [--loops]

VALU pipe cycles (MI355X_MICROARCH.md, cycle constants): plain 32-bit VALU 2
cycles per wave64 on a SIMD-32, packed-f32 (v_pk_*_f32) 4, transcendental 8.
"""
import collections
import re
import sys

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu_mov"
    if op.startswith(("v_add_co", "v_addc_co", "v_lshl_add_u64", "v_add_u32",
                      "v_lshlrev_b64", "v_add_lshl", "v_lshl_add", "v_mad_u",
                      "v_mad_i", "v_mul_lo", "v_mul_hi", "v_lshlrev_b32",
                      "v_and_b32", "v_or_b32", "v_lshl_or", "v_and_or",
                      "v_add3_u32", "v_ashr", "v_lshr", "v_sub_u32", "v_sub_co",
                      "v_subb", "v_bfe", "v_xor", "v_xad", "v_mbcnt",
                      "v_readfirstlane", "v_readlane", "v_writelane",
                      "v_cndmask", "v_cmp", "v_max_i", "v_min_i", "v_max_u",
                      "v_min_u", "v_add_i", "v_sub_i", "v_subrev_u",
                      "v_subrev_co", "v_or3", "v_perm", "v_alignbit",
                      "v_bfi", "v_not", "v_cvt")):
        return "valu_int"
    if op.startswith("v_"):
        return "valu_f32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
        return "vmem_rd"
    if op.startswith(("global_store", "flat_store", "buffer_store",
                      "scratch_store")):
        return "vmem_wr"
    if op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
        return "vmem_atomic"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


CYCLES = {"valu_pk": 4, "valu_trans": 8, "valu_mov": 2, "valu_int": 2,
          "valu_f32": 2}


def parse(path):
    """{kernel: [(label, op), ...]} for every .amdhsa kernel function."""
    kernels = {}
    cur = None
    label = ""
    func = re.compile(r"^([A-Za-z_.][\w$.]*):\s*(;.*)?$")
    inst = re.compile(r"^\s+([a-z][a-z0-9_]+)\b")
    for line in open(path):
        m = func.match(line)
        if m:
            name = m.group(1)
            if name.startswith(".L") or name.startswith("BB"):
                label = name
            else:
                cur = kernels.setdefault(name, [])
                label = ""
            continue
        if line.startswith("\t.") or line.startswith("."):
            if line.startswith("\t.end_amdhsa_kernel") or ".size" in line:
                pass
            continue
        m = inst.match(line)
        if m and cur is not None:
            cur.append((label, m.group(1)))
    return kernels


def demangle(names):
    import subprocess
    try:
        out = subprocess.run(["c++filt"] + names,
                             capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def table(rows):
    cols = ["valu_f32", "valu_pk", "valu_int", "valu_mov", "valu_trans", "lds",
            "vmem_rd", "vmem_wr", "vmem_atomic", "salu", "smem", "waitcnt",
            "nop", "barrier", "branch", "mfma", "other"]
    print("| kernel | total | " + " | ".join(cols) + " | VALU cycles |")
    print("|---|---|" + "---|" * (len(cols) + 1))
    for name, insts in rows:
        c = collections.Counter(classify(op) for _, op in insts)
        cyc = sum(CYCLES.get(k, 0) * v for k, v in c.items())
        print(f"| `{name}` | {len(insts)} | " +
              " | ".join(str(c.get(k, 0)) for k in cols) + f" | {cyc} |")


MINLEN = 8


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    loops = "--loops" in sys.argv
    path, pats = args[0], args[1:]
    kernels = parse(path)
    names = [n for n, v in kernels.items() if len(v) > 20]
    dm = demangle(names)
    rows = []
    for n in names:
        d = dm.get(n, n)
        if pats and not any(p in d for p in pats):
            continue
        short = re.sub(r"\(.*", "", d).replace("void ", "")
        if loops:
            by = collections.OrderedDict()
            for lab, op in kernels[n]:
                by.setdefault(lab, []).append((lab, op))
            for lab, ins in by.items():
                if len(ins) >= MINLEN:
                    rows.append((f"{short} {lab or 'entry'}", ins))
        else:
            rows.append((short, kernels[n]))
    table(rows)


if __name__ == "__main__":
    main()
