"""Builds libcastrec.so (HIP kernels + C ABI + native sampler) for gfx950 with hipcc, in-tree.

    python -m castrec_amd.build        (or: import castrec_amd.build; castrec_amd.build.build())
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcastrec.so")
LIB_TL = os.path.join(PKG, "libcastrec_tl.so")
SOURCES = ["cr_base.hip", "cr_embed.hip", "cr_layernorm.hip", "cr_eltwise.hip", "cr_gemm.hip",
           "cr_attn_fwd.hip", "cr_attn_bwd.hip", "cr_attn_bwd1.hip", "cr_attn_wide.hip", "cr_attn_bf.hip", "cr_gemm_bf.hip", "cr_block.hip", "cr_stack.hip", "cr_stack_bwd.hip", "cr_stack_bwd1.hip", "cr_wide.hip", "cr_head.hip", "cr_adam.hip", "cr_dist.hip", "cr_sampler.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wall", "-Wno-unused-function",
         # no SLP vectorisation: it packs adjacent scalar fp32 adds / multiplies into v_pk_*_f32.  (a) Beside MFMAs those cost more
         # than the two scalar ops (MI355X_MICROARCH.md, cycle constants).  (b) profiles/r04_flake/README.md: with SLP on, the last
         # statement of the LayerNorm backward in cr_stack_bwd1.hip becomes in-place v_pk_add -> v_pk_fma -> v_pk_mul chains on one
         # register pair, and in 1 replay of 5 .. 350 (schedule dependent; 111 recorded cases, all alike) the v_pk_fma's contribution
         # is missing in the low register of lanes 48..63; 0 of 3 000 on the same source without SLP.  Every pattern of the chain
         # is exact in isolation (seven probes, ~1e9 executions each): the trigger inside the kernel is not isolated, so the packing
         # itself is kept out.
         "-fno-slp-vectorize"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(obj, srcs):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False, timeline=False):
    """timeline=True builds libcastrec_tl.so with -DCR_TIMELINE=1: the per-wave phase stamps used by
    tools/block_ts.py and tools/attn_ts.py (compiled out of the production library)."""
    hipcc = _hipcc()
    bdir = "build_tl" if timeline else "build"
    lib = LIB_TL if timeline else LIB
    flags = FLAGS + (["-DCR_TIMELINE=1"] if timeline else []) + os.environ.get("CASTREC_EXTRA_FLAGS", "").split()
    # sources compiled WITH SLP vectorisation all the same: cr_attn_bf.hip -- its head-dim-32 forward is 37 % slower without the
    # packed fp32 arithmetic (config C4: 29.6 -> 40.6 us per launch); its 392 compiler-made v_pk_fma_f32 are all out-of-place (the
    # failing form above is the in-place chain: 47 of them in the SLP build of cr_stack_bwd1.hip, none here), and 4 000 replays of a
    # D = 128 / 4-head step hold the same bits (tools/diag_repro2.py, profiles/r04_flake/reproducibility_runs.json)
    slp_ok = os.environ.get("CASTREC_SLP_FILES", "cr_attn_bf.hip").split()
    os.makedirs(os.path.join(CSRC, bdir), exist_ok=True)
    headers = [os.path.join(CSRC, "cr_common.hpp"), os.path.join(CSRC, "cr_attn_common.hpp"), os.path.join(CSRC, "cr_bf16.hpp"), os.path.join(CSRC, "cr_rlayout.hpp"), os.path.join(CSRC, "cr_rbwd.hpp"),
               os.path.join(ROOT, "include", "castrec.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, bdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            fl = [f for f in flags if not (f == "-fno-slp-vectorize" and s in slp_ok)]
            cmd = [hipcc] + fl + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timeline="--timeline" in sys.argv))
