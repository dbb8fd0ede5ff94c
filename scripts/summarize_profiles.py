"""Condense rocprofv3 CSV output into the small tables kept under profiles/.

  python scripts/summarize_profiles.py trace <kernel_trace.csv> <steps> <out.csv> [warmup]   per (kernel, grid): launches, avg us, ms/step
  python scripts/summarize_profiles.py pmc <out.csv> <counter_collection.csv> [...]   per (kernel, counter): launches, mean per launch
"""
import collections, csv, sys


def short(name):
    return name if len(name) <= 180 else name[:180]


def trace(path, steps, out, warmup=0):
    """steps = steps summarised; warmup = steps skipped in front of them.  Round 6: the window is cut by TIME at optimizer steps - a
    GAN step ends with its second adam_kernel launch, so launches that start after the end of adam launch 2 * warmup and up to the end of
    adam launch 2 * (warmup + steps) are counted; everything else (first-use weight packing, the warm-up) is dropped.  (Rounds 1 - 5
    dropped the first warmup / (warmup + steps) of every group's launches BY COUNT: a kernel that runs only in the first step - the ~170
    lazy pack_wino4 launches - then showed up as ~20 launches 'per step'.)  Falls back to the by-count rule when the trace has no adam
    launches (pretrain-only or inference runs pass their own windows)."""
    rows_ = []
    adam_ends = []
    for r in csv.DictReader(open(path)):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        t0, t1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        rows_.append(((short(r["Kernel_Name"]), grid, wg), t0, (t1 - t0) / 1e3))
        if r["Kernel_Name"].startswith("adam_kernel") or r["Kernel_Name"].startswith("adam_dev_kernel"):
            adam_ends.append(t1)
    adam_ends.sort()
    groups = collections.OrderedDict()
    agg = {}
    if len(adam_ends) >= 2 * (warmup + steps):
        lo = adam_ends[2 * warmup - 1] if warmup else 0
        hi = adam_ends[2 * (warmup + steps) - 1]
        for key, t0, us in rows_:
            if lo < t0 <= hi:
                n, s = agg.get(key, (0, 0.0))
                agg[key] = (n + 1, s + us)
    else:
        for key, t0, us in rows_:
            groups.setdefault(key, []).append((t0, us))
        for key, lst in groups.items():
            lst.sort()
            drop = len(lst) * warmup // (warmup + steps) if len(lst) >= warmup + steps else 0
            kept = lst[drop:]
            agg[key] = (len(kept), sum(t for _, t in kept))
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "workgroup", f"launches({steps} steps)", "avg_us", "total_ms_per_step"])
        for (k, g, wg), (n, s) in rows:
            w.writerow([k, g, wg, n, round(s / n, 1), round(s / 1e3 / steps, 3)])


def pmc(out, paths):
    agg = collections.OrderedDict()
    for p in paths:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(p)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])          # summed over XCDs / instances
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (d, c), v in per_dispatch.items():
            k = (names[d], c)
            n, s = agg.get(k, (0, 0.0))
            agg[k] = (n + 1, s + v)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
        for (k, c), (n, s) in sorted(agg.items()):
            w.writerow([k, c, n, round(s / n, 1)])
    # side-car: hashes of the kernel sources the counters were taken on (bench.py refuses a summary whose kernel file changed since)
    import glob, hashlib, json, os
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pesr_amd", "csrc")
    meta = {"sources_sha256": {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()
                               for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))}}
    json.dump(meta, open(os.path.splitext(out)[0] + ".meta.json", "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    else:
        pmc(sys.argv[2], sys.argv[3:])
