#!/usr/bin/env python3
"""Per-position durations of the launches of a training step from a rocprofv3 kernel trace (…_kernel_trace.csv): the launches of
a step repeat in the same order, so position k of every step is the same launch (e.g. which of the four block backwards carries
the embedding scatter).   python3 tools/kt_by_position.py <dir with *_kernel_trace.csv> [launches per step]"""
import csv, glob, sys, statistics as st
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if not r["Kernel_Name"].startswith("at::") and "spin_kernel" not in r["Kernel_Name"] and not r["Kernel_Name"].startswith("__amd")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").split("(")[0][:28] for r in rows]
# period = distance between consecutive Adam launches
ad = [i for i, n in enumerate(names) if "adam" in n]
per = int(sys.argv[2]) if len(sys.argv) > 2 else st.mode([b - a for a, b in zip(ad, ad[1:])])
start = ad[len(ad) // 2] + 1
steps = []
i = start
while i + per <= len(rows) and "adam" in names[i + per - 1] or (i + per <= len(rows) and any("adam" in n for n in names[i:i + per])):
    steps.append(rows[i:i + per]); i += per
    if len(steps) >= 40: break
print("launches per step:", per, " steps used:", len(steps))
for k in range(per):
    d = [(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"])) / 1e3 for s in steps]
    gap = [(int(s[k]["Start_Timestamp"]) - int(s[k - 1]["End_Timestamp"])) / 1e3 for s in steps] if k else [0.0]
    nm = steps[0][k]["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    print("  %2d %-40s median %7.2f us  min %7.2f  gap before %5.2f us" % (k, nm, st.median(d), min(d), st.median(gap)))
tot = [(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3 for s in steps]
print("  step span (first start -> last end): median %.1f us" % st.median(tot))
