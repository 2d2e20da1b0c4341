"""Register / scratch / occupancy table of every kernel of one csrc/*.hip file
(hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled: no GPU needed).

    python tools/kernel_resources.py adjoint.hip [substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tike_amd", "csrc")


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-munsafe-fp-atomics", "-I../../include", "-I.",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:\[[^\]]*\] )?\s*(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|"
                      r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k.split(" ")[0]] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]],
                              capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", name).split("(")[0]
        if want in name:
            print(f"{name:70s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} scratch "
                  f"{r.get('ScratchSize'):>4s} occ {r.get('Occupancy'):>2s} LDS {r.get('LDS')}")


if __name__ == "__main__":
    main()
