#!/bin/bash
# rocprofv3 evidence for one command, into gpurun_out/<dir>: kernel trace + stats, then the three counter passes
# the bench's roofline block cites (separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# counters are never combined with the trace domains gpurun refuses).
#   tools/prof.sh <outdir under gpurun_out> [kt|pmc|sq|all] -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
set -o pipefail
out=gpurun_out/$1; what=$2; shift 3
mkdir -p "$out"
export TMPDIR=/tmp
run() { rocprofv3 "$@"; }
if [ "$what" = kt ] || [ "$what" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p_kt" -- "$@" > "$out/p_kt.log" 2>&1 || exit 1
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/p_fetch" -- "$@" > "$out/p_fetch.log" 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/p_write" -- "$@" > "$out/p_write.log" 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$out/p_mfma" -- "$@" > "$out/p_mfma.log" 2>&1 || exit 1
fi
# instruction mix and stall attribution (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots")
if [ "$what" = sq ] || [ "$what" = all ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d "$out/p_inst" -- "$@" > "$out/p_inst.log" 2>&1 || exit 1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/p_stall" -- "$@" > "$out/p_stall.log" 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES --output-format csv -d "$out/p_misc" -- "$@" > "$out/p_misc.log" 2>&1 || exit 1
fi
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/p_kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").split("(")[0]
        if n.startswith("at::") or n.startswith("__amd"):
            continue
        print("%-60s calls %5s avg %9.1f ns  min %8s max %8s" % (n[:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
