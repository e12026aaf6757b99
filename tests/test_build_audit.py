"""The build's ISA audit of the scan kernel's hand-counted load ring (comorag_amd/build.py: audit_ring) on small synthetic
listings: what it has to flag (a compiler copy of a ring register anywhere inside the ring loop — the textual order says nothing
there —, a reuse on a path that leaves the loop in front of a drain, scratch traffic) and what it has to let pass (the registers'
set-up in front of the loop wherever LLVM placed it in the text, reuse behind an `s_waitcnt vmcnt(0)`)."""
from comorag_amd.build import audit_ring

HEAD = "_Z11scan_kernelILi1ELi1ELi128ELi8ELi0ELi1ELi1EEv5ScanP:\n"
TAIL = ".Lfunc_end0:\n"


def _loads():
    # eight ring slots v[10:13] .. v[38:41], each waited for, consumed by an MFMA (A operand) and reloaded
    out = []
    for u in range(8):
        r = 10 + 4 * u
        out += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(7)", "\t;;#ASMEND",
                f"\tv_mfma_f32_32x32x16_bf16 v[100:115], v[{r}:{r + 3}], v[80:83], v[100:115]",
                "\t;;#ASMSTART", f"\tglobal_load_dwordx4 v[{r}:{r + 3}], v5, s[0:1] offset:0 nt", "\t;;#ASMEND"]
    return out


def _kernel(pre=(), in_loop=(), behind=(), cold=()):
    lines = ["\ts_load_dwordx2 s[0:1], s[4:5], 0x0"] + list(pre) + ["\ts_branch .LBB0_9", ".LBB0_1:                 ; =>This Loop Header: Depth=1"]
    lines += _loads() + list(in_loop) + ["\ts_cbranch_scc1 .LBB0_1", "; %bb.2:"] + list(behind) + ["\ts_endpgm"]
    lines += [".LBB0_9:"] + list(cold) + ["\ts_branch .LBB0_1"]
    return HEAD + "\n".join(lines) + "\n" + TAIL


KEY = (1, 1, 128, 8, 0)


def test_clean_ring_passes_and_setup_placed_behind_the_loop_is_not_held_against_it():
    asm = _kernel(cold=["\tv_mov_b32_e32 v10, 0", "\tv_mov_b32_e32 v41, 0"])       # zeroing the slots in front of the loop, laid out after it
    assert audit_ring(asm) == {KEY: True}


def test_copy_of_a_slot_at_the_loop_top_is_flagged_although_no_load_precedes_it_in_the_text():
    asm = _kernel()
    asm = asm.replace("; =>This Loop Header: Depth=1\n", "; =>This Loop Header: Depth=1\n\tv_mov_b64_e32 v[60:61], v[10:11]\n", 1)
    assert audit_ring(asm) == {KEY: False}


def test_mfma_may_read_a_slot_but_not_write_it():
    bad = _kernel(in_loop=["\tv_mfma_f32_32x32x16_bf16 v[10:25], v[80:83], v[84:87], v[10:25]"])
    assert audit_ring(bad) == {KEY: False}


def test_reuse_behind_the_loop_needs_a_drain_in_front_of_it():
    assert audit_ring(_kernel(behind=["\tv_mov_b32_e32 v12, 1"])) == {KEY: False}
    assert audit_ring(_kernel(behind=["\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\tv_mov_b32_e32 v12, 1"])) == {KEY: True}
    assert audit_ring(_kernel(behind=["\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND", "\tv_mov_b32_e32 v12, 1"])) == {KEY: True}
    # a branch around the drain is a path without one
    assert audit_ring(_kernel(behind=["\ts_cbranch_vccnz .LBB0_5", "\ts_waitcnt vmcnt(0)", ".LBB0_5:", "\tv_mov_b32_e32 v12, 1"])) == {KEY: False}


def test_cold_block_of_the_loop_laid_out_behind_it_counts_as_inside():
    asm = _kernel(in_loop=["\ts_cbranch_vccnz .LBB0_7"], behind=["\ts_waitcnt vmcnt(0)"])
    asm = asm.replace(TAIL, ".LBB0_7:                 ;   in Loop: Header=BB0_1 Depth=1\n\tv_mov_b32_e32 v20, v3\n\ts_branch .LBB0_1\n" + TAIL)
    assert audit_ring(asm) == {KEY: False}


def test_scratch_traffic_fails_the_variant():
    assert audit_ring(_kernel(pre=["\tscratch_store_dwordx2 off, v[2:3], off"])) == {KEY: False}


def test_wrong_number_of_ring_registers_fails():
    asm = _kernel().replace("global_load_dwordx4 v[38:41]", "global_load_dwordx4 v[34:37]")
    assert audit_ring(asm) == {KEY: False}


def test_valu_written_sgpr_in_front_of_an_asm_load_is_flagged_and_five_wait_states_clear_it():
    """build.audit_asm_sgpr_hazard: hipcc brings a spilled SGPR back with v_readlane and pads no wait states in front of an inline-asm
    vector-memory instruction that reads it as its address (round 6: the READY hint of the finishing stage faulted on exactly that)."""
    from comorag_amd.build import audit_asm_sgpr_hazard
    bad = """_Z4kernv:
	v_readlane_b32 s12, v136, 16
	v_readlane_b32 s13, v136, 17
	;;#ASMSTART
	global_load_dword v121, v97, s[12:13] offset:0x80 sc1
	;;#ASMEND
	s_endpgm
"""
    hits = audit_asm_sgpr_hazard(bad)
    assert len(hits) == 1 and hits[0][0] == "_Z4kernv" and "v_readlane_b32 s13" in hits[0][2]
    good = bad.replace("\tglobal_load_dword", "\ts_nop 4\n\tglobal_load_dword")
    assert audit_asm_sgpr_hazard(good) == []
    # compiler-issued loads (outside an asm statement) are hipcc's own business; four plain instructions + the nop-less load are one state short
    assert audit_asm_sgpr_hazard(bad.replace(";;#ASMSTART\n", "").replace(";;#ASMEND\n", "")) == []
    spaced = bad.replace("\t;;#ASMSTART", "\ts_mov_b32 s1, 0\n\ts_mov_b32 s2, 0\n\ts_mov_b32 s3, 0\n\ts_mov_b32 s4, 0\n\t;;#ASMSTART")
    assert len(audit_asm_sgpr_hazard(spaced)) == 1
    assert audit_asm_sgpr_hazard(spaced.replace("s_mov_b32 s4, 0", "s_mov_b32 s4, 0\n\ts_mov_b32 s5, 0")) == []
