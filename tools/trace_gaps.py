"""Where the GPU idles: gaps between consecutive kernels of a rocprofv3
--kernel-trace CSV, over the last `epochs` epochs of a bench run.

    python tools/trace_gaps.py <kernel_trace.csv> [marker-kernel-substring] [epochs]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "resident_kernel"
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    S = [int(r["Start_Timestamp"]) for r in rows]
    E = [int(r["End_Timestamp"]) for r in rows]
    idx = [i for i, n in enumerate(names) if marker in n]
    per_epoch = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    a, b = idx[-per_epoch * epochs - 1], idx[-1]
    busy = sum(E[i] - S[i] for i in range(a, b))
    span = S[b] - S[a]
    print(f"window {span / 1e6:.3f} ms ({epochs} epochs), kernels busy "
          f"{busy / 1e6:.3f} ms = {busy / span:.4f}, launches {b - a}")
    gaps = collections.defaultdict(lambda: [0, 0])
    big = []
    for i in range(a, b):
        g = S[i + 1] - E[i]
        key = (names[i][:48], names[i + 1][:48])
        gaps[key][0] += 1
        gaps[key][1] += g
        if g > 20000:
            big.append((g, names[i][:60], names[i + 1][:60]))
    tot = sum(v[1] for v in gaps.values())
    print(f"total gap {tot / 1e6:.3f} ms; gaps > 20 us: {len(big)} "
          f"totalling {sum(g for g, _, _ in big) / 1e6:.3f} ms")
    for g, x, y in sorted(big, reverse=True)[:12]:
        print(f"  {g / 1e3:9.1f} us  {x} -> {y}")
    print("by pair:")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {v[1] / 1e3:9.1f} us n={v[0]:4d} avg {v[1] / v[0] / 1e3:7.2f}  "
              f"{k[0]} -> {k[1]}")
    # per-kernel busy time inside the window
    t = collections.defaultdict(lambda: [0, 0])
    for i in range(a, b):
        t[names[i][:70]][0] += 1
        t[names[i][:70]][1] += E[i] - S[i]
    print("busy by kernel (per epoch):")
    for k, v in sorted(t.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {v[1] / 1e6 / epochs:8.3f} ms n={v[0] / epochs:6.1f} avg "
              f"{v[1] / v[0] / 1e3:8.1f} us  {k}")


if __name__ == "__main__":
    main()
