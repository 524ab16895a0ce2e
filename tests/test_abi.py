"""The C-ABI library loads and exports every symbol include/castrec.h declares (no compute calls)."""
import ctypes
import os
import re

import castrec_amd  # noqa: F401
from castrec_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "castrec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cr_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) >= 20
    so = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(so, n), "libcastrec.so does not export %s" % n
    assert sorted(L.EXPORTS) == names


def test_version_and_error_string_without_gpu():
    assert L.lib.cr_version() >= 100
    assert isinstance(L.lib.cr_last_error(), bytes)
    # argument validation happens before any HIP call: a NULL descriptor is rejected with a message
    rc = L.lib.cr_adam_step(None, None)
    assert rc == -1 and b"cr_adam_step" in L.lib.cr_last_error()


def test_round3_entries_validate_their_arguments_without_gpu():
    """cr_stack_block_bwd / cr_rows_pack / cr_rows_add reject NULL and unsupported descriptions before any HIP call."""
    bd, ad, ext = L.BlockBwdDesc(), L.AttnDesc(), L.BlockBwd1Ext()
    assert L.lib.cr_stack_block_bwd_supported(ctypes.byref(bd), ctypes.byref(ad), 4, 20, L.PREC_BF16X3) == 0     # D = 0
    rc = L.lib.cr_stack_block_bwd(ctypes.byref(bd), ctypes.byref(ad), ctypes.byref(ext), None, None, 4, 20, L.PREC_BF16X3, None)
    assert rc == -1 and b"cr_stack_block_bwd" in L.lib.cr_last_error()
    bd.f.D, bd.f.M, bd.n_slabs = 64, 80, 4                     # D = 64: the bias gradients need a spare column (D < 64)
    assert L.lib.cr_stack_block_bwd_supported(ctypes.byref(bd), ctypes.byref(ad), 4, 20, L.PREC_BF16X3) == 0
    bd.f.D = 50
    assert L.lib.cr_stack_block_bwd_supported(ctypes.byref(bd), ctypes.byref(ad), 4, 20, L.PREC_F32) == 0      # bf16 arithmetic only
    assert L.lib.cr_rows_pack(None, None, 1, 1, 1, None, None, None, 1, None) == -1 and b"cr_rows_pack" in L.lib.cr_last_error()
    assert L.lib.cr_rows_add(None, None, 1, 1, 1, None) == -1 and b"cr_rows_add" in L.lib.cr_last_error()
    assert L.lib.cr_ids_ring_next(None, 2, 8, None, None, None) == -1 and b"cr_ids_ring_next" in L.lib.cr_last_error()
    buf = (ctypes.c_int32 * 64)()
    p = ctypes.addressof(buf) + (-ctypes.addressof(buf)) % 16
    assert L.lib.cr_ids_ring_next(p, 0, 6, p + 64, p, None) == -1 and b"n_slots" in L.lib.cr_last_error()
    # the id ring inside the Adam launch: not together with the row-sparse update (it reads the step's ids)
    ad = L.AdamDesc()
    ad.p = ad.m = ad.v = ad.state = ad.table_grad = p
    ad.n_table, ad.lazy_ids, ad.lazy_flags, ad.n_lazy_ids, ad.lazy_rows, ad.lazy_D = 8, p, p, 4, 2, 4
    ad.ids_ring, ad.ids_ring_slots, ad.ids_slot_elems, ad.ids_dst = p, 2, 8, p + 64
    assert L.lib.cr_adam_step(ctypes.byref(ad), None) == -1 and b"exclude each other" in L.lib.cr_last_error()


def test_round5_entries_validate_their_arguments_without_gpu():
    """cr_batch_index_layout / cr_index_build / cr_table_grad / cr_adam_desc.tg reject bad descriptions before any HIP call."""
    lay = L.IndexLayout()
    assert L.lib.cr_batch_index_layout(0, 10, 0, 32, 16, ctypes.byref(lay)) == -1 and b"cr_batch_index_layout" in L.lib.cr_last_error()
    assert L.lib.cr_batch_index_layout(12, 10, 5, 32, 16, ctypes.byref(lay)) == -1            # T_pos must divide M
    assert L.lib.cr_batch_index_layout(12, 10, 4, 65, 16, ctypes.byref(lay)) == -1            # at most 64 lane groups
    assert L.lib.cr_batch_index_layout(12, 10, 4, 64, 8, ctypes.byref(lay)) == 0              # the plan of D = 48: 64 groups of 8 occurrences
    assert lay.cap_occ == 48 and lay.bitmap_words == 1 and lay.off_recs == 8 and lay.total_words % 4 == 0 and lay.cap_blocks >= 2
    assert L.lib.cr_index_build(None, None, None, None, None) == -1 and b"cr_index_build" in L.lib.cr_last_error()
    g = L.TgradDesc()
    assert L.lib.cr_table_grad(ctypes.byref(g), None, None) == -1 and b"cr_table_grad" in L.lib.cr_last_error()
    buf = (ctypes.c_int32 * 64)()
    p = ctypes.addressof(buf) + (-ctypes.addressof(buf)) % 16
    g.lay, g.D, g.rows, g.seq_emb, g.coef, g.ld_rows, g.ld_emb, g.index, g.part_rows, g.tickets = lay, 67, p, p, p, 67, 67, p, p, p
    assert L.lib.cr_table_grad(ctypes.byref(g), p, None) == -1 and b"hidden size" in L.lib.cr_last_error()     # odd and above 64
    g.D, g.ld_rows, g.ld_emb = 48, 50, 48
    assert L.lib.cr_table_grad(ctypes.byref(g), p, None) == -1 and b"leading dimensions" in L.lib.cr_last_error()
    g.ld_rows = 48
    g.index = None
    assert L.lib.cr_table_grad(ctypes.byref(g), p, None) == -1 and b"index" in L.lib.cr_last_error()
    g.index = p
    ad = L.AdamDesc()
    ad.p = ad.m = ad.v = ad.state = p
    ad.n_table, ad.tg = 8, ctypes.pointer(g)                 # (V + T_pos) * D = 14 * 48 is what the index describes
    assert L.lib.cr_adam_step(ctypes.byref(ad), None) == -1 and b"n_table must be" in L.lib.cr_last_error()
    ad.n_table = 14 * 48
    ad.lazy_ids, ad.lazy_flags, ad.n_lazy_ids, ad.lazy_rows, ad.lazy_D = p, p, 4, 2, 4
    assert L.lib.cr_adam_step(ctypes.byref(ad), None) == -1 and b"exclude each other" in L.lib.cr_last_error()


def test_struct_sizes_match_c_layout(tmp_path):
    """sizeof of every descriptor as gcc lays it out from include/castrec.h == the ctypes mirror."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    pairs = [("cr_rng", L.Rng), ("cr_embed_desc", L.EmbedDesc), ("cr_embed_bwd_desc", L.EmbedBwdDesc), ("cr_ln_desc", L.LnDesc),
             ("cr_ln_bwd_desc", L.LnBwdDesc), ("cr_gemm_desc", L.GemmDesc), ("cr_wgrad_desc", L.WgradDesc),
             ("cr_elt_desc", L.EltDesc), ("cr_attn_desc", L.AttnDesc), ("cr_attn_bwd_desc", L.AttnBwdDesc),
             ("cr_block_desc", L.BlockDesc), ("cr_block_bwd_desc", L.BlockBwdDesc), ("cr_block_bwd1_ext", L.BlockBwd1Ext), ("cr_block_tail_desc", L.BlockTailDesc),
             ("cr_stack_desc", L.StackDesc), ("cr_head_desc", L.HeadDesc), ("cr_adam_desc", L.AdamDesc),
             ("cr_index_layout", L.IndexLayout), ("cr_tgrad_desc", L.TgradDesc)]
    # every structure the header declares has a mirror in this list
    hdr = open(os.path.join(ROOT, "include", "castrec.h")).read()
    declared = set(re.findall(r"\}\s*(cr_[a-z_0-9]+)\s*;", hdr))
    assert declared == {c for c, _ in pairs}, declared ^ {c for c, _ in pairs}
    # sizeof AND the offset of every field (same field names on both sides)
    probes = []
    for cname, ct in pairs:
        probes.append(("sizeof(%s)" % cname, ctypes.sizeof(ct), cname))
        for fname, _ in ct._fields_:
            probes.append(("offsetof(%s, %s)" % (cname, fname), getattr(ct, fname).offset, cname + "." + fname))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "castrec.h"\nint main(void){' +
                   "".join('printf("%%zu\\n", (size_t)%s);' % e for e, _, _ in probes) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert len(got) == len(probes)
    for (_, want, what), n in zip(probes, got):
        assert want == n, (what, want, n)
