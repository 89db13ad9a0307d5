"""ISA-level invariants of the built gfx950 code object, checked on the CPU by disassembling libmatfact_hip.so
(llvm-objdump / llvm-readelf of the ROCm image; tools/isa.py).  Bit-exactness against matFact.c rests on structural
facts of the machine code that no numerical test can see until they break rarely:

  * no fused multiply-add where the reference multiplies and adds separately (matFact.c:45-51, mat2d.c:133-137);
  * nothing spilled to scratch in the kernels on the timed path;
  * the matrix-core recommendation really is on v_mfma_f64_16x16x4_f64;
  * the hand-written regions keep their invariants BY CONSTRUCTION: an LDS read hipcc cannot see (inline asm) is waited
    for before anything touches its destination -- round 2's one-wrong-sum-per-1e8-blocks was a compiler copy (v_mov_b64)
    of such a pending destination in front of its s_waitcnt, DESIGN.md 5.2c -- , no LDS read is pending while the
    v_fmac_f64_dpp chain of the ordered sums executes, and M0 is written right in front of every LDS-DMA instruction
    that reads it.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa  # noqa: E402

pytestmark = pytest.mark.skipif(not isa.have_tools() or not os.path.exists(isa.DEFAULT_LIB),
                                reason="needs llvm-objdump/llvm-readelf/c++filt and the built library")


@pytest.fixture(scope="module")
def kernels():
    k = {n: b for n, b in isa.disassemble().items() if "rocprim" not in n}
    assert len(k) > 60, "kernel symbols not found in the code object"
    return k


@pytest.fixture(scope="module")
def meta():
    return {n: m for n, m in isa.metadata().items() if "rocprim" not in n}


def _ops(body):
    return [isa.split(i)[0] for i in body]


# kernels whose sums must be formed exactly as the serial program forms them: separate multiply and add
EXACT = ("sweep_kernel", "sweep_dma_kernel", "sweep_coop_kernel", "sweep_resident_kernel", "stream_resident_kernel",
         "recommend_kernel", "predict_kernel")


def test_no_fused_multiply_add_in_the_exact_kernels(kernels):
    checked = 0
    for name, body in kernels.items():
        if not any("mf::" + e in name for e in EXACT):
            continue
        checked += 1
        fused = [i for i in body if re.match(r"v_(fma|fmac|mad|pk_fma)_f64", i)]
        assert not fused, (name, fused[:3])
        ops = set(_ops(body))
        assert "v_mul_f64" in ops and "v_add_f64" in ops, name
    assert checked >= 60


def test_ordered_sums_fuse_only_the_exact_product_by_one(kernels):
    """v_fmac_f64_dpp acc, v, 1.0: the only FP64 DPP form gfx950 has; the product by 1.0 is exact, the rounding is the
    add's.  Every one of them must take the SAME register pair as its multiplier (the 1.0), and nothing else fuses."""
    dpp = kernels["void mf::ordered_sum_kernel<true>(mf::OrderedSumArgs)"]
    plain = kernels["void mf::ordered_sum_kernel<false>(mf::OrderedSumArgs)"]
    fm = [i for i in dpp if i.startswith("v_fmac_f64_dpp")]
    assert len(fm) >= 3 * 2 * 31, len(fm)   # three depth classes x (full block + partial block) of 16 / 15 entries x 2 columns
    ones = {isa.split(i)[1].split(",")[2].split()[0] for i in fm}
    assert len(ones) == 1, ones
    one_regs = isa.regs(next(iter(ones)))
    for i in dpp:   # the 1.0 is never written after its initialisation inside the loop nest: only v_mov of the constant
        op, rest = isa.split(i)
        if op.startswith("v_") and isa.regs(rest.split(",")[0]) & one_regs and not op.startswith("v_fmac_f64_dpp"):
            assert op in ("v_mov_b32_e32", "v_mov_b64_e32"), i
    assert all("row_newbcast:" in i and "row_mask:0xf" in i and "bank_mask:0xf" in i for i in fm)
    assert not [i for i in dpp if re.match(r"v_(fma|fmac)_f64", i) and "_dpp" not in i]
    assert not [i for i in plain if re.match(r"v_(fma|fmac)_f64", i)]
    assert "v_add_f64" in _ops(plain)


def _walk_pending_lds_reads(body, name):
    """every inline-asm style LDS read (ds_read_b128) is followed, in straight-line code, by s_waitcnt lgkmcnt(0) before any
    instruction names a register of its destination; returns the number of reads checked"""
    n = 0
    for pos, ins in enumerate(body):
        op, rest = isa.split(ins)
        if op != "ds_read_b128":
            continue
        dest = isa.regs(rest.split(",")[0])
        assert len(dest) == 4, ins
        for nxt in body[pos + 1:]:
            nop, nrest = isa.split(nxt)
            if nop == "s_waitcnt" and "lgkmcnt(0)" in nrest:
                break
            assert not nop.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")), (name, ins, "control flow before the wait", nxt)
            assert not (isa.regs(nrest) & dest), (name, ins, "destination touched before lgkmcnt(0)", nxt)
        else:
            raise AssertionError((name, ins, "no lgkmcnt(0) after the read"))
        n += 1
    return n


def test_ordered_sum_lds_reads_are_waited_for_before_their_destination_is_touched(kernels):
    """The invariant whose violation was round 2's rare wrong sum: in tools/micro/alt/libmatfact_hip_base.so this walk
    finds `v_mov_b64 v[6:7], v[24:25]` between `ds_read_b128 v[24:27]` and its `s_waitcnt lgkmcnt(0)`."""
    for form in ("true", "false"):
        name = "void mf::ordered_sum_kernel<%s>(mf::OrderedSumArgs)" % form
        assert _walk_pending_lds_reads(kernels[name], name) >= 9   # 3 depth classes x (seed, steady state, tail)


def test_pipelined_step_of_the_ordered_sums_is_one_closed_statement(kernels):
    """The steady state: ds_read_b128 of block b+1, the 32 v_fmac_f64_dpp of block b, the transfer of block b+D-1 and the
    s_waitcnt lgkmcnt(0), back to back -- one asm statement, nothing of hipcc's in between (two per depth class: the
    registers swap roles, plus the odd step).  The read's destination is never a chain operand of the same statement."""
    body = kernels["void mf::ordered_sum_kernel<true>(mf::OrderedSumArgs)"]
    steps = 0
    for pos, ins in enumerate(body):
        if not ins.startswith("ds_read_b128") or pos + 37 >= len(body) or not body[pos + 1].startswith("v_fmac_f64_dpp"):
            continue
        dest = isa.regs(isa.split(ins)[1].split(",")[0])
        chain = body[pos + 1:pos + 33]
        assert all(c.startswith("v_fmac_f64_dpp") for c in chain), chain
        assert ["row_newbcast:%d " % (e // 2) in c + " " for e, c in enumerate(chain)] == [True] * 32
        assert not any(isa.regs(isa.split(c)[1]) & dest for c in chain), (ins, "chain reads the pending quad")
        assert body[pos + 33].startswith("s_mov_b32 m0,") and body[pos + 34].startswith("s_nop")
        assert body[pos + 35].startswith("global_load_lds_dwordx4") and body[pos + 36] == "s_waitcnt lgkmcnt(0)", body[pos + 33:pos + 37]
        assert body[pos - 1].startswith("s_waitcnt vmcnt("), body[pos - 1]
        steps += 1
    assert steps == 9, steps


def test_m0_is_written_in_front_of_every_lds_dma_of_the_ordered_sums(kernels):
    """M0 (the LDS destination of global_load_lds) is a reserved register hipcc rewrites at will: the asm statement that
    issues the transfer writes it itself -- s_mov_b32 m0, sX; s_nop 0; global_load_lds_dwordx4."""
    for form in ("true", "false"):
        body = kernels["void mf::ordered_sum_kernel<%s>(mf::OrderedSumArgs)" % form]
        n = 0
        for pos, ins in enumerate(body):
            if ins.startswith("global_load_lds_dwordx4"):
                assert body[pos - 1].startswith("s_nop"), body[pos - 3:pos + 1]
                assert body[pos - 2].startswith("s_mov_b32 m0,"), body[pos - 3:pos + 1]
                n += 1
        assert n >= 6


def test_hand_counted_vmcnt_of_the_ordered_sums(kernels):
    """Depth class D keeps D-1 blocks issued ahead.  The pipelined step reads block b+1 while blocks up to b+D-2 are issued:
    at most D-3 newer transfers may be outstanding (32 / 16 / 8 -> 29 / 13 / 5).  The plain form reads block b itself:
    D-2 (30 / 14 / 6)."""
    def depths(body, follow):
        seen = set()
        for pos, ins in enumerate(body):
            if ins.startswith("ds_read_b128") and body[pos - 1].startswith("s_waitcnt vmcnt(") and follow(body, pos):
                seen.add(int(re.search(r"vmcnt\((\d+)\)", body[pos - 1]).group(1)))
        return seen
    dpp = kernels["void mf::ordered_sum_kernel<true>(mf::OrderedSumArgs)"]
    assert depths(dpp, lambda b, p: b[p + 1].startswith("v_fmac_f64_dpp")) == {29, 13, 5}
    plain = kernels["void mf::ordered_sum_kernel<false>(mf::OrderedSumArgs)"]
    waits = {int(m.group(1)) for i in plain for m in [re.match(r"s_waitcnt vmcnt\((\d+)\)$", i)] if m}
    assert {30, 14, 6} <= waits, waits


def test_timed_path_kernels_do_not_spill(kernels, meta):
    # K = 128 at two workgroups per CU sits at the 256-register limit (128 VGPRs of L operand, 64 accumulators): hipcc parks
    # ONE value of the prologue in scratch and fetches it after the last tile -- nothing between the first and the last
    # matrix instruction may touch scratch (a scratch access inside the loop would also join the hand-counted vmcnt)
    allowed = {"void mf::recommend_mfma2_kernel<8, 4, 2, 4>(mf::RecMfmaArgs)": 16}
    for name, body in kernels.items():
        if "mf::" not in name:
            continue
        m = meta[name]
        if name in allowed:
            mf = [i for i, x in enumerate(body) if x.startswith("v_mfma")]
            assert not [x for x in body[mf[0]:mf[-1] + 1] if x.startswith("scratch_")], name
            assert m[".private_segment_fixed_size"] <= allowed[name] and m[".vgpr_spill_count"] <= 2, (name, m)
            continue
        assert not [i for i in body if i.startswith("scratch_")], name
        assert m[".private_segment_fixed_size"] == 0 and m[".vgpr_spill_count"] == 0, (name, m)


def test_recommendation_runs_on_the_fp64_matrix_cores(kernels):
    forms = [n for n in kernels if "recommend_mfma_kernel" in n]
    assert len(forms) >= 6
    for n in forms:
        mf = [i for i in kernels[n] if i.startswith("v_mfma_f64_16x16x4")]
        assert len(mf) >= 48, (n, len(mf))
    assert not [i for i in kernels["mf::recommend_kernel(mf::RecArgs)"] if i.startswith("v_mfma")]


def test_two_per_cu_recommend_keeps_its_matrix_stream_gapless(kernels, meta):
    """recommend_mfma2_kernel<NC, QC, TU, WAVES> (K = 4 QC NC): 4 TU QC NC matrix instructions per tile, the L operand in registers (no spill,
    at most 256 VGPRs so that two workgroups share a CU), R fragments by ds_read2st64_b64 with immediate offsets only --
    at most one vector add per chunk in the whole k-loop -- , a cheap reject of exactly 32 compares whose lane masks go to
    scalar registers, every fragment read issued IN FRONT of the matrix instructions of the k-step before its own (the
    scheduler sinks them otherwise), R chunks by LDS-DMA with a scalar base, M0 written in front of each."""
    shapes = [(nc, 5, 2, 4) for nc in (1, 2, 3, 4, 5)] + [(nc, 4, 2, 4) for nc in (1, 2, 3, 4, 5, 6, 7, 8)] + [(8, 8, 1, 8)]   # K = 20 NC, 16 NC, 256
    for nc, qc, tu, waves in shapes:
        name = "void mf::recommend_mfma2_kernel<%d, %d, %d, %d>(mf::RecMfmaArgs)" % (nc, qc, tu, waves)
        body, m = kernels[name], meta[name]
        assert m[".vgpr_count"] <= 256 and m[".vgpr_spill_count"] <= (2 if (nc, qc) == (8, 4) else 0), (name, m)   # K=128: see the spill test
        ring = 3 * 2 * qc * 128 * 16
        assert m[".group_segment_fixed_size"] + ring <= (80 if waves == 4 else 160) * 1024        # two workgroups per CU by LDS (one of eight waves)
        mf = [i for i, x in enumerate(body) if x.startswith("v_mfma_f64_16x16x4")]
        assert len(mf) == 4 * tu * qc * nc, (name, len(mf))
        assert sum(1 for x in body[mf[0]:mf[0] + 4 * tu] if x.startswith("v_mfma") and x.rstrip().endswith(", 0")) >= 1   # C = 0, no zeroing
        loop = body[mf[0]:mf[-1] + 1]
        assert not [x for x in loop if x.startswith(("ds_read_b64", "ds_read2_b64 "))], name
        assert 2 * qc * nc - 2 <= sum(1 for x in loop if x.startswith("ds_read2st64_b64")) <= 2 * qc * nc   # the first fragment of a tile is read under the tile before
        valu = [x for x in loop if x.startswith("v_") and not x.startswith("v_mfma")]
        # per chunk one base-address add; the out-of-line transfer issue holds none (scalar base, per-tile row offset)
        assert len(valu) <= 2 * nc + 12, (name, valu)
        for pos, x in enumerate(body):
            if x.startswith("global_load_lds_dwordx4"):
                assert re.search(r", s\[\d+:\d+\]", x), x
                assert body[pos - 1].startswith("s_nop") and body[pos - 2].startswith("s_mov_b32 m0,"), body[pos - 3:pos + 1]
        tail = body[mf[-1] + 1:]
        tail = tail[:next(i for i, x in enumerate(tail) if x.startswith("s_cbranch"))]   # up to the branch on "norms rule out non-finite scores"
        assert [x.split()[0] for x in tail if x.startswith("v_")] == ["v_cmp_nle_f64_e32", "v_cmp_nle_f64_e64", "v_cmp_nle_f64_e32", "v_cmp_nle_f64_e32"] * (4 * tu), tail
    # k-steps 2.. of every chunk: the two reads of the NEXT fragment stand before the first matrix instruction of the step
    body = kernels["void mf::recommend_mfma2_kernel<5, 5, 2, 4>(mf::RecMfmaArgs)"]
    mf = [i for i, x in enumerate(body) if x.startswith("v_mfma_f64_16x16x4")]
    groups = [mf[g:g + 8] for g in range(0, len(mf), 8)]
    ahead = 0
    for g, nxt in zip(groups, groups[1:]):
        reads = [i for i in range(g[0] - 12, nxt[0]) if body[i].startswith("ds_read2st64_b64")]
        ahead += sum(1 for i in reads if i < g[-1])
    assert ahead >= 2 * (len(groups) - 1) - 4, ahead


def test_sweeps_gather_by_lds_dma(kernels):
    """every LDS-DMA form of the sweep moves its rows with global_load_lds_dwordx4 (no VGPR staging, no ds_write of the tile)"""
    n = 0
    for name, body in kernels.items():
        if "mf::sweep_dma_kernel" in name or "mf::sweep_coop_kernel" in name:
            assert any(i.startswith("global_load_lds_dwordx4") for i in body), name
            n += 1
    assert n >= 30


def test_occupancy_budget_of_the_single_wave_kernels(meta):
    """one-wave workgroups rely on many resident workgroups per CU: the production sweeps stay within 128 VGPRs (4 waves per
    SIMD by registers; LDS is the tighter bound) and the ordered sums within 64"""
    for name, m in meta.items():
        if re.search(r"mf::sweep_dma_kernel<(100|128|256), \d, 0>", name):
            assert m[".vgpr_count"] <= 128, (name, m[".vgpr_count"])
    assert meta["void mf::ordered_sum_kernel<true>(mf::OrderedSumArgs)"][".vgpr_count"] <= 64
