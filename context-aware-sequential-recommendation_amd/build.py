"""Builds libcastrec.so (HIP kernels + C ABI + native sampler) for gfx950 with hipcc, in-tree.

    python -m castrec_amd.build        (or: import castrec_amd.build; castrec_amd.build.build())
"""
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcastrec.so")
LIB_TL = os.path.join(PKG, "libcastrec_tl.so")
SOURCES = ["cr_base.hip", "cr_embed.hip", "cr_layernorm.hip", "cr_eltwise.hip", "cr_gemm.hip",
           "cr_attn_fwd.hip", "cr_attn_bwd.hip", "cr_attn_bwd1.hip", "cr_attn_wide.hip", "cr_attn_bf.hip", "cr_gemm_bf.hip", "cr_block.hip", "cr_stack.hip", "cr_stack_bwd.hip", "cr_stack_bwd1.hip", "cr_wide.hip", "cr_head.hip", "cr_adam.hip", "cr_tgrad.hip", "cr_dist.hip", "cr_sampler.cpp", "cr_index.cpp"]
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


# ---- the ISA check -----------------------------------------------------------------------------------------------------
# The sources whose device code is scanned at build time (every register-layout kernel, the wide row kernels and the one file
# that IS compiled with SLP vectorisation).  What is refused: a v_pk_fma_f32 that accumulates in place (destination pair ==
# SrcC pair) on a register pair that one of the two preceding vector instructions wrote with a packed fp32 op -- the
# pk_add -> pk_fma -> pk_mul chain on one pair that profiles/r04_flake/README.md pins the round-3 wrong-dx episodes to (47 of them
# in an SLP build of cr_stack_bwd1.hip, none in production).  cr_common.hpp's cr_ln_bwd_tail keeps the source from forming it;
# this keeps any other statement, flag or compiler version from forming it unseen.
ISA_CHECKED = ["cr_stack.hip", "cr_stack_bwd.hip", "cr_stack_bwd1.hip", "cr_wide.hip", "cr_attn_bf.hip",
               # (round 5: every source with matrix instructions -- the two-shape rule; each holds ONE shape today)
               "cr_block.hip", "cr_attn_fwd.hip", "cr_attn_bwd.hip", "cr_attn_bwd1.hip", "cr_gemm.hip", "cr_gemm_bf.hip"]
SLP_PROTECTED = ("cr_stack", "cr_wide")          # file-name prefixes that may never be compiled with SLP vectorisation
_PK_F32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")
_LABEL = re.compile(r"^([A-Za-z_$][\w$.]*):")
_MODIFIER = re.compile(r"\s(op_sel|op_sel_hi|neg_lo|neg_hi|clamp|mul:|div:|quad_perm|row_|bank_mask|bound_ctrl|dst_sel|dst_unused|src\d_sel)")


def _operands(rest):
    m = _MODIFIER.search(" " + rest)
    if m:
        rest = (" " + rest)[:m.start()]
    return [o.strip().lstrip("-").strip("|") for o in rest.split(",") if o.strip()]


_VREG = re.compile(r"^v(?:\[(\d+):(\d+)\]|(\d+))$")
_WRITES_VGPR = ("v_", "ds_read", "global_load", "buffer_load", "flat_load", "scratch_load")


def _vregs(op):
    """The VGPR numbers an operand names: 'v[4:7]' -> {4, 5, 6, 7}, 'v9' -> {9}, anything else -> empty."""
    m = _VREG.match(op)
    if not m:
        return frozenset()
    if m.group(3) is not None:
        return frozenset([int(m.group(3))])
    return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))


def scan_isa(text):
    """Scans an AMDGPU assembly listing.  Returns dict(kernels, pk_fma, pk_add, pk_mul, in_place_any, mfma, mixed_mfma_chains,
    violations) where a violation is (function, line number, the chain's lines): see ISA_CHECKED.  Two kinds:
    * an in-place packed fp32 chain (profiles/r04_flake/README.md);
    * an MFMA whose SrcC is the destination of the latest MFMA that wrote those registers and that one has ANOTHER shape (round 5:
      v_mfma_f32_16x16x16_bf16 accumulating onto v_mfma_f32_16x16x32_bf16's result gave wrong, run-dependent values in the first
      two accumulator registers -- the plain-bf16 block backward's weight gradients at the headline length; one shape along an
      accumulation chain is the form that is exact: cr_rbwd.hpp wgrad_accum).  Listing order, not control flow: a heuristic, which
      is what a build-time tripwire can be."""
    out = dict(kernels=0, pk_fma=0, pk_add=0, pk_mul=0, in_place_any=0, mfma={}, mixed_mfma_chains=0, violations=[])
    func, last = "?", []                                  # last: the two preceding vector instructions (mnemonic, dst, line)
    mfma_dst = {}                                         # register set -> (mnemonic, line) of the MFMA that wrote it last
    for no, line in enumerate(text.splitlines(), 1):
        if line and not line[0].isspace():
            m = _LABEL.match(line)
            if m and not line.startswith("."):            # a function's entry label (basic-block labels start with .LBB)
                func, last, mfma_dst = m.group(1), [], {}
                out["kernels"] += 1
            continue
        t = line.strip()
        if not t.startswith(_WRITES_VGPR):
            continue
        mn, _, rest = t.partition(" ")
        ops = _operands(rest.split(";")[0])
        dst = ops[0] if ops else ""
        dregs = _vregs(dst)
        if mn.startswith("v_mfma"):
            out["mfma"][mn] = out["mfma"].get(mn, 0) + 1
            srcc = _vregs(ops[3]) if len(ops) >= 4 else frozenset()
            prev = mfma_dst.get(srcc)
            if prev and prev[0] != mn:
                out["mixed_mfma_chains"] += 1
                out["violations"].append((func, no, [prev[1], t]))
            for k in [k for k in mfma_dst if k & dregs]:
                del mfma_dst[k]
            if dregs:
                mfma_dst[dregs] = (mn, t)
        elif dregs and mfma_dst:
            for k in [k for k in mfma_dst if k & dregs]:
                del mfma_dst[k]
        if not t.startswith("v_"):
            continue
        if mn in _PK_F32:
            out[mn[2:-4]] += 1
            if dst in ops[1:]:
                out["in_place_any"] += 1
            if mn == "v_pk_fma_f32" and len(ops) >= 4 and ops[3] == dst:
                for pm, pd, pl in last:
                    if pm in _PK_F32 and pd == dst:
                        out["violations"].append((func, no, [pl, t]))
                        break
        last = (last + [(mn, dst, t)])[-2:]
    return out


def _source_digest(src, headers, flags):
    h = hashlib.sha256()
    for f in [src] + headers:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def isa_summary_path(source_name, timeline=False):
    return os.path.join(CSRC, "build_tl" if timeline else "build", source_name + ".isa.json")


def check_isa_file(source_name, flags, headers, hipcc=None, listing=None):
    """Compiles one source to its gfx950 assembly with exactly `flags` (or reads `listing`), scans it, writes the summary beside
    the objects and raises on a violation."""
    src = os.path.join(CSRC, source_name)
    if listing is None:
        listing = os.path.join(CSRC, "build", source_name + ".s")
        cmd = [hipcc or _hipcc()] + flags + ["-x", "hip", "-S", "--cuda-device-only", src, "-o", listing]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s" % (" ".join(cmd), r.stderr))
    with open(listing) as fh:
        res = scan_isa(fh.read())
    res["source"], res["digest"], res["flags"] = source_name, _source_digest(src, headers, flags), flags
    with open(isa_summary_path(source_name), "w") as fh:
        json.dump(res, fh, indent=1)
    if res["violations"]:
        f, no, chain = res["violations"][0]
        raise RuntimeError("%s: %d forbidden instruction chains in the device code (in-place packed fp32, or MFMAs of two shapes on one accumulator; first: %s, line %d: %s) -- see build.py ISA_CHECKED"
                           % (source_name, len(res["violations"]), f, no, " ; ".join(chain)))
    return res


def isa_is_current(source_name, flags, headers):
    """True when the ISA summary beside the objects was made from the present source, headers and flags, and is clean."""
    try:
        with open(isa_summary_path(source_name)) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return False
    return d.get("digest") == _source_digest(os.path.join(CSRC, source_name), headers, flags) and not d.get("violations")


def file_flags(source_name, timeline=False):
    """The exact compile flags of one source (what build() uses and what the ISA check repeats)."""
    extra = os.environ.get("CASTREC_EXTRA_FLAGS", "").split()
    slp_ok = os.environ.get("CASTREC_SLP_FILES", "cr_attn_bf.hip").split()
    # correctness of the register-layout kernels must not hang on an environment variable: SLP vectorisation cannot be switched
    # back on for them, neither per file nor through the extra flags
    bad = [f for f in slp_ok if f.startswith(SLP_PROTECTED)]
    if bad:
        raise RuntimeError("CASTREC_SLP_FILES names %s: these sources are never compiled with SLP vectorisation "
                           "(profiles/r04_flake/README.md)" % bad)
    if any(f in ("-fslp-vectorize", "-fvectorize-slp") or f.startswith("-fslp-vectorize") for f in extra):
        raise RuntimeError("CASTREC_EXTRA_FLAGS may not re-enable SLP vectorisation (profiles/r04_flake/README.md)")
    flags = FLAGS + (["-DCR_TIMELINE=1"] if timeline else []) + extra
    return [f for f in flags if not (f == "-fno-slp-vectorize" and source_name in slp_ok)]


# every object depends on these (a change rebuilds the library; the ISA summaries carry their digest)
HEADERS = [os.path.join(CSRC, h) for h in ("cr_common.hpp", "cr_attn_common.hpp", "cr_bf16.hpp", "cr_rlayout.hpp", "cr_rbwd.hpp")] \
    + [os.path.join(ROOT, "include", "castrec.h")]
# ... and single sources on these besides
EXTRA_DEPS = {"cr_adam.hip": ["cr_tgrad.hpp"], "cr_tgrad.hip": ["cr_tgrad.hpp"]}


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
    # (sources compiled WITH SLP vectorisation all the same -- file_flags: cr_attn_bf.hip.  Its head-dim-32 forward is 37 % slower
    # without the packed fp32 arithmetic (config C4: 29.6 -> 40.6 us per launch); its compiler-made v_pk_fma_f32 never accumulate in
    # place behind a packed write of the same pair -- checked on every build, below -- and 4 000 replays of a D = 128 / 4-head step
    # hold the same bits: tools/diag_repro2.py, profiles/r04_flake/reproducibility_runs.json)
    os.makedirs(os.path.join(CSRC, bdir), exist_ok=True)
    headers = HEADERS
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, bdir, s + ".o")
        objs.append(obj)
        fl = file_flags(s, timeline)
        checked = s in ISA_CHECKED and not timeline
        deps = [src] + headers + [os.path.join(CSRC, h) for h in EXTRA_DEPS.get(s, [])]
        if checked and not force and not _stale(obj, deps) and not isa_is_current(s, fl, headers):
            force_this = True                              # an object without a current ISA summary (older build, other flags)
        else:
            force_this = False
        if force or force_this or _stale(obj, deps):
            cmd = [hipcc] + fl + (["-save-temps=obj"] if checked else []) + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append((cmd, s if checked else None, fl))

    def run(job):
        cmd, checked_src, fl = job if isinstance(job, tuple) else (job, None, None)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        if checked_src:
            # the device listing -save-temps left beside the object: scanned (a violation fails the build), then the temporaries go
            stem = checked_src[:-4]
            bd = os.path.join(CSRC, bdir)
            try:
                check_isa_file(checked_src, fl, headers, listing=os.path.join(bd, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s"))
            finally:
                for f in glob.glob(os.path.join(bd, stem + "-hip-amdgcn-*")) + glob.glob(os.path.join(bd, stem + "-host-*")) \
                        + glob.glob(os.path.join(bd, stem + ".hip-hip-*")):
                    os.remove(f)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"], None, None))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timeline="--timeline" in sys.argv))
