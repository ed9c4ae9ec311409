"""CPU regression guard over the gfx950 ISA of libmoca_hip.so (no GPU: the library is cross-compiled, `llvm-objdump -d`
and the code-object metadata are read by tools/isa_report.py).

DESIGN 4.1 names three compiler behaviours that each cost more than any kernel-structure choice and are easy to re-introduce by an
innocent edit: hipcc drains the LDS-DMA stream (`s_waitcnt vmcnt(0)`) (1) before an LDS access it cannot prove disjoint from a DMA in
flight and (2) before a write of a register it knows as a load destination; (3) register-allocator copies / spills around inline-asm
waits.  None of them changes a result -- only a GPU timing run would show them.  Here they show as: a `vmcnt(0)` inside the MFMA span
of a main loop, a changed MFMA / DMA / fragment-read count per loop body, VGPR spills or scratch, or a register count above the
occupancy the design assumes (two waves per SIMD for the 8-wave GEMM blocks = 256 VGPRs, four for attention_v4 = 128)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def isa():
    import isa_report
    from moca_video_amd import lib
    if not os.path.exists(os.path.join(isa_report.LLVM, "llvm-objdump")) or shutil.which("c++filt") is None:
        pytest.skip("llvm-objdump / c++filt not available")
    r = isa_report.analyse(lib.LIB_PATH)
    assert len(r) > 80, f"only {len(r)} kernels found in {lib.LIB_PATH}"
    return r


# kernel -> (MFMAs, LDS-DMA instructions, LDS fragment reads, barriers) of ONE main-loop body, as designed:
#   staggered 8-wave kernels: a k-tile PAIR per body -- 2 x 25 MFMAs of an 80 x 80 wave tile (2 x 32 of 64 x 128, 2 x 30 of 80 x 96),
#   8 DMA instructions (A + W of both k-tiles), (5 + 5) x 2 fragment reads (8 + 4 / 5 + 6 per k-tile on the other shapes), 4 barriers
#   (LOAD / MFMA segments of the two halves); 4-wave two-blocks-per-CU kernels: 64 MFMAs of a 128 x 128 wave tile pair, 12 DMA;
#   256-row split-K kernel: one k-tile per body, 32 (BN 128) / 40 (BN 160) MFMAs, 6 / 7 DMA instructions, ONE barrier.
MAIN_LOOPS = {
    **{f"gemm_w80s_kernel<{m}, {s}>": (50, 8, 20, 4) for m in range(4) for s in (0, 1)},
    "gemm_w80s_kernel<0, 2>": (64, 8, 24, 4),
    "gemm_w80s_kernel<0, 3>": (60, 8, 22, 4),
    "gemm_sqp_kernel<true>": (64, 8, 24, 4),
    "gemm_sqp_kernel<false>": (64, 8, 24, 4),
    "gemm_g4p_kernel<true>": (64, 12, 24, 2),
    "gemm_g4p_kernel<false>": (64, 12, 24, 2),
    **{f"gemm_glds_kernel<128, {m}, {f}>": (32, 6, 16, 1) for m in range(3) for f in ("true", "false")},
    **{f"gemm_glds_kernel<160, {m}, {f}>": (40, 7, 18, 1) for m in range(3) for f in ("true", "false")},
    # weight-stationary 320 -> 320 kernel (8 waves, K halves): per 32-row strip a wave issues 2 row tiles x 5 k-steps x 5 column tiles of MFMAs
    # with W in registers, its share of the strip's LDS-DMA pieces (5 with a residual, 2-3 without: the conditional third is outside the
    # loop body the report isolates), 10 fragment reads + 5 exchange reads (+ 3 residual reads)
    **{f"gemm_ws_kernel<false, {e}>": (50, 2, 15, None) for e in range(3)},
    **{f"gemm_ws_kernel<true, {e}>": (50, 5, 18, 2) for e in range(3)},
}


def test_main_loops_have_the_designed_instruction_counts_and_no_dma_drain(isa):
    for name, (mfma, dma, dsrd, barr) in MAIN_LOOPS.items():
        assert name in isa, f"{name} is not in the library"
        lp = isa[name]["loop"]
        assert lp is not None, f"{name}: no MFMA loop found"
        got = (lp["mfma"], lp["lds_dma"], lp["ds_read"], lp["barrier"] if barr is not None else None)
        assert got == (mfma, dma, dsrd, barr), f"{name}: main loop (MFMA, LDS-DMA, ds_read, barriers) = {got}, designed {(mfma, dma, dsrd, barr)}"
        assert lp["vmcnt0"] == 0, f"{name}: {lp['vmcnt0']} `s_waitcnt vmcnt(0)` in the main loop ({lp['vmcnt0_inside']} between its MFMAs): " \
                                  "the DMA stream is drained every iteration (DESIGN 4.1)"
        if not name.startswith("gemm_ws_"):           # (its partial-sum exchange is 5 ds_write_b128 per strip by design)
            assert lp["ds_write"] == 0, f"{name}: LDS writes in the main loop: {lp}"
        assert lp["global_load"] == 0, f"{name}: register-staged traffic in the main loop: {lp}"


# the plain / `+res` flavour of the persistent kernel is NOT in the default dispatch (MOCA_TUNE_GEMM_SQP=2 only): hipcc
# spills two VGPRs of its epilogue (3 scratch instructions, none in the main loop)
KNOWN_SPILLS = {"gemm_sqp_kernel<false>": 2}


def test_no_kernel_spills_vector_registers_or_uses_scratch(isa):
    for name, d in isa.items():
        allowed = KNOWN_SPILLS.get(name, 0)
        assert d.get("vgpr_spill_count", 0) <= allowed, f"{name}: {d['vgpr_spill_count']} VGPR spills"
        if not allowed:
            assert d.get("private_segment_fixed_size", 0) == 0 and d["scratch"] == 0, f"{name}: scratch memory in use"
        if d["loop"] is not None:
            assert d["loop"]["scratch"] == 0, f"{name}: scratch traffic inside the main loop"


def test_register_counts_keep_the_designed_occupancy(isa):
    """8-wave blocks (w80s, sqp, glds) and the 4-wave kernels that run two blocks per CU hold two waves per SIMD: <= 256 registers
    (VGPR + AGPR, unified file of 512 per SIMD lane); attention_v4 runs four waves per SIMD: <= 128; no AGPR use in the GEMMs (an
    accumulator that moves to AGPRs costs v_accvgpr moves in every epilogue)."""
    for name, d in isa.items():
        regs = d.get("vgpr_count", 0) + d.get("agpr_count", 0)
        if name.startswith("gemm_"):
            assert regs <= 256, f"{name}: {regs} registers: fewer than two waves per SIMD"
            assert d.get("agpr_count", 0) == 0, f"{name}: accumulators moved to AGPRs"
        if "attention_v4" in name:
            assert regs <= 128, f"{name}: {regs} registers: fewer than four waves per SIMD"
            assert d["loop"]["mfma"] >= 16                                      # 16 MFMAs per (32 queries x 64 keys) tile


def test_every_gemm_kernel_of_the_dispatch_is_present(isa):
    for stem, n in (("gemm_w80s_kernel", 10), ("gemm_glds_kernel", 12), ("gemm_g4_kernel", 6), ("gemm_sqp_kernel", 2),
                    ("gemm_g4p_kernel", 2), ("gemm_f16_kernel", 6)):
        have = [k for k in isa if k.startswith(stem + "<")]
        assert len(have) == n, f"{stem}: {len(have)} instantiations, expected {n}: {have}"
