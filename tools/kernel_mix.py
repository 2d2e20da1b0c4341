"""Static instruction mix of one kernel of a csrc/*.hip file, per basic block
(cross-compiled, no GPU): python tools/kernel_mix.py ptycho.hip <mangled-name substring> [min VALU]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, want = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = f"/tmp/kmix_{os.path.basename(src)}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                "-munsafe-fp-atomics", "-I../../include", "-I.", "-S", "--cuda-device-only", src,
                "-o", out], cwd=os.path.join(ROOT, "tike_amd", "csrc"), check=True,
               capture_output=True)
s = open(out).read()
names = sorted(set(re.findall(r"^(_Z\w+):", s, flags=re.M)))
for name in [n for n in names if want in n]:
    i = s.index(name + ":")
    j = s.index(".amdhsa_kernel " + name)
    body = s[i:j].splitlines()
    print(name, "lines", len(body))
    blocks, cur = [], []
    for l in body:
        if re.match(r"^\.LBB", l):
            blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    for b in blocks:
        v = [l.split()[0] for l in b if re.match(r"\s+v_", l)]
        if len(v) < floor:
            continue
        sc = sum(bool(re.match(r"\s+s_", l)) for l in b)
        ld = sum("global_load" in l for l in b)
        st = sum("global_store" in l for l in b)
        ds = sum(bool(re.match(r"\s+ds_", l)) for l in b)
        scr = sum("scratch_" in l for l in b)
        print(f"  {b[0][:24]:24s} VALU {len(v):4d} SALU {sc:4d} gld {ld:3d} gst {st:3d} ds {ds:3d} scratch {scr:3d}")
        print("     ", collections.Counter(v).most_common(10))
