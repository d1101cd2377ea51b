#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into a short markdown table (library kernels only).

usage: summarize_rocprof.py <dir with *_kernel_stats.csv / *_counter_collection.csv> <out.md> [title]
"""
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def warm_averages(src):
    """kernel name -> average duration (ns) of all its launches but the first, from the kernel
    trace: the first launch of a kernel pays for loading its code object and cold caches, so the
    plain average of a short run overstates the steady state."""
    durs = {}
    for path in sorted(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)):
        rows = list(csv.DictReader(open(path)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            durs.setdefault(r["Kernel_Name"], []).append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v[1:]) / (len(v) - 1) for k, v in durs.items() if len(v) > 1}


def main():
    src, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out)
    lines = ["# %s" % title, ""]
    warm = warm_averages(src)
    for path in sorted(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)):
        rows = list(csv.DictReader(open(path)))
        total = sum(float(r["TotalDurationNs"]) for r in rows)
        lines += ["## kernel stats (`rocprofv3 --kernel-trace --stats`), %s" % os.path.basename(path), "",
                  "warm avg = all launches but each kernel's first (from the kernel trace).", "",
                  "| kernel | calls | avg ms | warm avg ms | min ms | max ms | % of GPU time |",
                  "|---|---|---|---|---|---|---|"]
        other = 0.0
        for r in rows:
            if "ic::" in r["Name"]:
                w = warm.get(r["Name"])
                lines.append("| `%s` | %s | %.4f | %s | %.4f | %.4f | %.2f |" % (
                    short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                    "%.4f" % (w / 1e6) if w is not None else "-", float(r["MinNs"]) / 1e6,
                    float(r["MaxNs"]) / 1e6, 100.0 * float(r["TotalDurationNs"]) / total))
            else:
                other += float(r["TotalDurationNs"])
        lines.append("| (torch kernels generating the synthetic inputs, copies) | | | | | | %.2f |" % (100.0 * other / total))
        lines.append("")
    for path in sorted(glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True)):
        agg = {}
        for r in csv.DictReader(open(path)):
            if "ic::" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        lines += ["## counters (`rocprofv3 --pmc`), %s" % os.path.basename(path), "",
                  "| kernel | counter | dispatches | mean per dispatch |", "|---|---|---|---|"]
        for (k, c), (n, v) in sorted(agg.items()):
            lines.append("| `%s` | %s | %d | %.6g |" % (k, c, n, v / n))
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
