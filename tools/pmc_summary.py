#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel trace --stats + separate --pmc passes) into profiles/<tag>_*.

    python tools/pmc_summary.py gpurun_out r01_e

HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes read -- calibrated here on k_ln_fwd, whose read is known
(M*D*4 B = 5000 KiB at the headline shape and it reports ~2520 KiB) -- so traffic = (2*FETCH + WRITE) KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def newest(files, window=600.0):
    """gpurun merges every call's outputs into the same directory: only the files of the latest run of a pass count"""
    files = sorted(files, key=os.path.getmtime)
    return [f for f in files if os.path.getmtime(files[-1]) - os.path.getmtime(f) <= window] if files else []


def agg(pattern, counter):
    d = collections.defaultdict(list)
    for f in newest(glob.glob(pattern)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


stats = newest(glob.glob(os.path.join(src, "p_kt", "*", "*_kernel_stats.csv")))[-1:]
if stats:
    shutil.copy(stats[0], os.path.join(out, tag + "_kernel_stats.csv"))
fe = agg(os.path.join(src, "p_fetch", "*", "*_counter_collection.csv"), "FETCH_SIZE")
wr = agg(os.path.join(src, "p_write", "*", "*_counter_collection.csv"), "WRITE_SIZE")
mf = {c: agg(os.path.join(src, "p_mfma", "*", "*_counter_collection.csv"), c)
      for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES")}
SQ_PASSES = {"p_inst": ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                        "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES"),
             "p_stall": ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                         "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
             "p_misc": ("SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_IFETCH", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_ACTIVE_INST_SCA",
                        "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CU_CYCLES")}
sq = {c: agg(os.path.join(src, d, "*", "*_counter_collection.csv"), c) for d, cs in SQ_PASSES.items() for c in cs}
dur = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
summary = {}
for k in fe:
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    f, w = fe[k][0], wr.get(k, (0.0, 0))[0]
    e = dict(fetch_size_kib=round(f, 1), write_size_kib=round(w, 1), hbm_traffic_bytes=int((2 * f + w) * 1024))
    if k in dur:
        e["avg_ns"], e["calls"] = round(dur[k][0], 1), dur[k][1]
    for c in mf:
        if k in mf[c]:
            e[c] = round(mf[c][k][0])
    if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over all 1024 SIMDs
        e["mfma_busy_frac"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    # instruction mix and stall shares (tools/prof.sh sq): per-launch sums over all waves.  SQ_WAVE_CYCLES / SQ_WAIT_* /
    # SQ_ACTIVE_INST_* count quad-cycles of wave lifetime (MI355X_MICROARCH.md cycle-constants table): shares of the
    # wave-cycles are what a wave spends parked (s_waitcnt / barrier), stalled at issue, or issuing
    for c in sq:
        if k in sq[c]:
            e[c] = round(sq[c][k][0])
    if e.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round((e.get("SQ_INSTS_VALU", 0) - e["SQ_INSTS_MFMA"]) / e["SQ_INSTS_MFMA"], 2)   # SQ_INSTS_VALU counts MFMAs too
        e["lds_per_mfma"] = round(e.get("SQ_INSTS_LDS", 0) / e["SQ_INSTS_MFMA"], 2)
        e["salu_per_mfma"] = round(e.get("SQ_INSTS_SALU", 0) / e["SQ_INSTS_MFMA"], 2)
    wc = e.get("SQ_WAVE_CYCLES")
    if wc and "SQ_WAIT_ANY" in e:
        e["share_wait_any"] = round(e["SQ_WAIT_ANY"] / wc, 4)
        e["share_wait_inst_any"] = round(e["SQ_WAIT_INST_ANY"] / wc, 4)
        e["share_wait_inst_lds"] = round(e["SQ_WAIT_INST_LDS"] / wc, 4)
        e["share_active_inst_any"] = round(e["SQ_ACTIVE_INST_ANY"] / wc, 4)
        e["share_active_inst_valu"] = round(e["SQ_ACTIVE_INST_VALU"] / wc, 4)
    if e.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"], 4)
    summary[k] = e
with open(os.path.join(out, tag + "_pmc_summary.json"), "w") as fjson:
    json.dump(summary, fjson, indent=1, sort_keys=True)
for k, e in sorted(summary.items(), key=lambda kv: -kv[1].get("avg_ns", 0) * kv[1].get("calls", 0)):
    print("%-34s avg %8.1f us  traffic %7.2f MB  mfma_busy %s  valu/mfma %s  lds/mfma %s  wait_any %s  wait_inst %s (lds %s)  issuing %s  lds_conflict %s"
          % (k[:34], e.get("avg_ns", 0) / 1e3, e["hbm_traffic_bytes"] / 1e6, e.get("mfma_busy_frac"), e.get("valu_per_mfma"), e.get("lds_per_mfma"),
             e.get("share_wait_any"), e.get("share_wait_inst_any"), e.get("share_wait_inst_lds"), e.get("share_active_inst_any"),
             e.get("lds_bank_conflict_share")))
