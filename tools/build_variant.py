#!/usr/bin/env python3
"""Build a variant of libtike_amd.so for a same-box A/B (tools/ab_libs.sh).

    python tools/build_variant.py <name> [<file> <old text> <new text>] ...

Copies tike_amd/csrc to a scratch directory, applies the textual replacements
(each must match), builds, and leaves tools/probe/_lib/lib_<name>.so (git- and
history-ignored, but it travels to the GPU box with the snapshot).  `base` with
no replacements is the current source.  Then, on the GPU box:

    bash tools/ab_libs.sh 2 base <name> ...        # WL=c5 STEPS=4 for another workload
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tike_amd", "csrc")
WORK = os.environ.get("TIKE_VARIANT_DIR", "/tmp/tike_variant")


def main():
    name, args = sys.argv[1], sys.argv[2:]
    assert len(args) % 3 == 0, __doc__
    os.makedirs(WORK, exist_ok=True)
    for f in (glob.glob(os.path.join(SRC, "*.hip")) + glob.glob(
            os.path.join(SRC, "*.h")) + glob.glob(os.path.join(SRC, "*.cpp"))):
        text = open(f).read().replace('"../../include/tike_amd.h"',
                                      '"tike_amd.h"')
        open(os.path.join(WORK, os.path.basename(f)), "w").write(text)
    mk = open(os.path.join(SRC, "Makefile")).read()
    mk = mk.replace("-I../../include", f"-I{ROOT}/include").replace(
        "../../include/tike_amd.h", f"{ROOT}/include/tike_amd.h")
    open(os.path.join(WORK, "Makefile"), "w").write(mk)
    for i in range(0, len(args), 3):
        path = os.path.join(WORK, args[i])
        text = open(path).read()
        assert args[i + 1] in text, f"no match in {args[i]}: {args[i + 1][:60]}"
        open(path, "w").write(text.replace(args[i + 1], args[i + 2], 1))
    out = subprocess.run(["make", "-C", WORK, "-B", "-j6"], capture_output=True,
                         text=True)
    if out.returncode:
        sys.exit(out.stdout[-3000:] + out.stderr[-3000:])
    dst = os.path.join(ROOT, "tools", "probe", "_lib")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(WORK, "libtike_amd.so"),
                os.path.join(dst, f"lib_{name}.so"))
    print("built", os.path.join(dst, f"lib_{name}.so"))


if __name__ == "__main__":
    main()
