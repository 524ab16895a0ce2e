"""The device code of the register-layout kernels holds no in-place packed fp32 chain and no accumulation chain of two MFMA shapes
(castrec_amd/build.py: ISA_CHECKED).

profiles/r04_flake/README.md pins round 3's wrong-dx episodes to ONE compiler-made form: v_pk_add_f32 t ; v_pk_fma_f32 t, a, b, t ;
v_pk_mul_f32 out, c, t on one register pair (SLP vectorisation of the LayerNorm backward's last statement).  The source can no
longer form it (cr_common.hpp cr_ln_bwd_tail: three scalar instructions in inline assembly), the build scans every listing for
it and refuses to link, build.py refuses the environment overrides that would bring SLP back for those files -- and this test,
which needs no GPU, checks all three."""
import json
import os

import pytest

import castrec_amd  # noqa: F401
from castrec_amd import build as B

# modelled on the failing revision's listing (profiles/r04_flake: .LBB0_385 of k_stack_block_bwd, SLP build)
BAD = """
\t.text
_Z17k_stack_block_bwdILb1ELi50ELi1EEv8B1Args:                 ; @_Z17k_stack_block_bwdILb1ELi50ELi1EEv8B1Args
\ts_waitcnt lgkmcnt(0)
\tv_pk_add_f32 v[150:151], v[150:151], v[92:93] neg_lo:[0,1] neg_hi:[0,1]
\ts_nop 0
\tv_pk_fma_f32 v[150:151], v[164:165], v[132:133], v[150:151] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]
\tv_pk_mul_f32 v[150:151], v[94:95], v[150:151] op_sel_hi:[0,1]
"""
GOOD = """
_Z17k_stack_block_bwdILb1ELi50ELi1EEv8B1Args:
\tv_pk_add_f32 v[150:151], v[150:151], v[92:93]
\tv_add_f32_e32 v1, v2, v3
\tv_mul_f32_e32 v4, v2, v3
\tv_pk_fma_f32 v[150:151], v[164:165], v[132:133], v[150:151]
\tv_pk_fma_f32 v[28:29], v[28:29], s[94:95], v[34:35]
\tv_sub_f32 v7, v8, v9
\tv_fma_f32 v7, -v10, v11, v7
\tv_mul_f32 v7, v12, v7
"""


def test_scanner_sees_the_failing_chain_and_nothing_else():
    bad = B.scan_isa(BAD)
    assert bad["kernels"] == 1 and len(bad["violations"]) == 1
    f, line, chain = bad["violations"][0]
    assert f.startswith("_Z17k_stack_block_bwd") and "v_pk_fma_f32 v[150:151]" in chain[1] and chain[0].startswith("v_pk_add_f32")
    good = B.scan_isa(GOOD)              # the packed write is three vector instructions back; an in-place SrcA is not the form
    assert good["violations"] == [] and good["pk_fma"] == 2 and good["in_place_any"] >= 2


# round 5: the plain-bf16 block backward at the headline length (listing of k_stack_block_bwd<false, 50, 1> before the fix): the odd
# seventh tile's K = 16 product accumulates onto the K = 32 product of tiles 4 / 5 -- registers 0 / 1 of that accumulator came out
# wrong, run-dependent (tools/probes/bf16_w2_blocks.py)
MIXED = """
_Z17k_stack_block_bwdILb0ELi50ELi1EEv6B1Args:
\tds_read_b64_tr_b16 v[104:105], v106 offset:22528
\tds_read_b64_tr_b16 v[106:107], v106 offset:24576
\tv_mfma_f32_16x16x32_bf16 v[104:107], v[100:103], v[104:107], v[116:119]
\tv_mfma_f32_16x16x32_bf16 v[100:103], v[100:103], v[108:111], v[120:123]
\ts_cmp_lt_i32 s87, 7
.LBB5_253:
\tv_lshl_add_u32 v92, v92, 1, 0
\tds_read_b64_tr_b16 v[108:109], v94 offset:12288
\tv_mfma_f32_16x16x16_bf16 v[104:107], v[108:109], v[94:95], v[104:107]
\tv_mfma_f32_16x16x16_bf16 v[100:103], v[108:109], v[110:111], v[100:103]
"""
ONE_SHAPE = """
_Z17k_stack_block_bwdILb0ELi50ELi1EEv6B1Args:
\tv_mfma_f32_16x16x32_bf16 v[104:107], v[100:103], v[104:107], v[116:119]
\tv_mfma_f32_16x16x32_bf16 v[104:107], v[108:111], v[92:95], v[104:107]
\tv_mfma_f32_16x16x32_bf16 v[100:103], v[100:103], v[108:111], v[120:123]
\tv_mov_b32_e32 v100, 0
\tv_mfma_f32_16x16x16_bf16 v[100:103], v[108:109], v[110:111], v[100:103]
"""


def test_scanner_sees_an_accumulation_chain_of_two_mfma_shapes():
    bad = B.scan_isa(MIXED)
    assert bad["mixed_mfma_chains"] == 2 and len(bad["violations"]) == 2
    assert bad["mfma"] == {"v_mfma_f32_16x16x32_bf16": 2, "v_mfma_f32_16x16x16_bf16": 2}
    assert "16x16x32" in bad["violations"][0][2][0] and "16x16x16" in bad["violations"][0][2][1]
    ok = B.scan_isa(ONE_SHAPE)           # one shape along a chain; a K = 16 product on registers a vector instruction rewrote is no chain
    assert ok["violations"] == [] and ok["mixed_mfma_chains"] == 0


def _headers():
    return B.HEADERS


@pytest.mark.parametrize("src", B.ISA_CHECKED)
def test_production_listing_is_clean(src):
    """The summary the build wrote beside the object must come from the PRESENT source, headers and flags (digest) and hold no
    violation; a stale or missing one is made now by compiling the file to its listing with build.py's exact flags."""
    flags = B.file_flags(src)
    if not B.isa_is_current(src, flags, _headers()):
        B.check_isa_file(src, flags, _headers())                                   # hipcc -S --cuda-device-only; raises on a violation
    with open(B.isa_summary_path(src)) as fh:
        d = json.load(fh)
    assert d["violations"] == [] and d["kernels"] >= 1 and d["mixed_mfma_chains"] == 0
    assert len(d["mfma"]) == 1, d["mfma"]                             # every source holds ONE matrix instruction shape (round 5)
    if src.startswith(("cr_stack", "cr_wide", "cr_attn_bf", "cr_gemm_bf")):
        assert set(d["mfma"]) == {"v_mfma_f32_16x16x32_bf16"}
    assert ("-fno-slp-vectorize" in d["flags"]) == (src != "cr_attn_bf.hip")
    if src != "cr_attn_bf.hip":
        # without SLP every packed op comes from explicit two-element vector code (the hi / lo splits): adds and multiplies of
        # conversions -- the LayerNorm tail is scalar by construction
        assert "-fno-slp-vectorize" in flags


def test_build_refuses_to_bring_slp_back(monkeypatch):
    monkeypatch.setenv("CASTREC_SLP_FILES", "cr_attn_bf.hip cr_stack_bwd1.hip")
    with pytest.raises(RuntimeError, match="never compiled with SLP"):
        B.file_flags("cr_stack_bwd1.hip")
    monkeypatch.setenv("CASTREC_SLP_FILES", "cr_wide.hip")
    with pytest.raises(RuntimeError, match="never compiled with SLP"):
        B.file_flags("cr_adam.hip")
    monkeypatch.delenv("CASTREC_SLP_FILES")
    monkeypatch.setenv("CASTREC_EXTRA_FLAGS", "-DX=1 -fslp-vectorize")
    with pytest.raises(RuntimeError, match="SLP"):
        B.file_flags("cr_stack.hip")
    monkeypatch.setenv("CASTREC_EXTRA_FLAGS", "-DX=1")
    assert "-fno-slp-vectorize" in B.file_flags("cr_stack.hip") and "-DX=1" in B.file_flags("cr_stack.hip")
    assert "-fno-slp-vectorize" not in B.file_flags("cr_attn_bf.hip")


def test_the_layernorm_tail_is_inline_assembly():
    """cr_rbwd.hpp / cr_wide.hip route the statement through cr_ln_bwd_tail; nothing spells it in C++ any more."""
    for f in ("cr_rbwd.hpp", "cr_wide.hip"):
        text = open(os.path.join(B.CSRC, f)).read()
        assert "cr_ln_bwd_tail(" in text and "- c1 - xc[ct][r] * c2" not in text and "- c1 - h * c2" not in text
    common = open(os.path.join(B.CSRC, "cr_common.hpp")).read()
    assert "v_sub_f32 %0, %1, %2" in common and "v_fma_f32 %0, -%3, %4, %0" in common and "v_mul_f32 %0, %5, %0" in common
