"""CPU-side check of the shipped device code: every gfx950 kernel of libgpx.so is read back from the code-object
metadata (tests/codeobj.py) and must use no scratch memory and spill no vector registers.  Scratch is what turned the
first GEMM core into a 79 TFLOP/s kernel and spilled SGPRs sat on the serial pivot chain of the
diagonal-block kernel, so both are kept out by a test rather than by inspection."""
import os

import pytest

import codeobj

# SGPR spills go to VGPR lanes (v_writelane / v_readlane), not to memory: tolerated where they sit outside the hot recurrences.
# diag_ldlm_kernel<double>: 21 lane-group masks of its 32 elimination steps, parked at entry.  small_alpha_kernel (second launch
# of the small-model create): ~20 pointers of its argument block live across its three phases, two of them go to VGPR lanes.
# wide_factor_kernel<double>: the 128 x 128 dataflow tile with that diagonal-block routine inlined (its 21 + the tile's own
# kernel arguments, parked across the routine).
SGPR_SPILL_LIMIT = {"diag_ldlm_kernelId": 24, "small_alpha_kernel": 4, "wide_factor_kernelId": 48}


@pytest.fixture(scope="module")
def kernels(gpx, tmp_path_factory):
    ks = codeobj.kernels(gpx.LIB_PATH, tmp_path_factory.mktemp("codeobj"))
    assert len(ks) > 50, "could not read the kernels of libgpx.so"
    return ks


def test_no_kernel_uses_scratch(kernels):
    bad = [(k["name"], k["private_segment_fixed_size"]) for k in kernels if k["private_segment_fixed_size"] != 0]
    assert not bad, "kernels with a private segment (scratch): %s" % bad


def test_no_kernel_spills_vector_registers(kernels):
    bad = [(k["name"], k["vgpr_spill_count"]) for k in kernels if k["vgpr_spill_count"] != 0]
    assert not bad, "kernels with VGPR spills: %s" % bad


def test_no_kernel_spills_scalar_registers(kernels):
    def limit(name):
        return max([v for key, v in SGPR_SPILL_LIMIT.items() if key in name] + [0])

    bad = [(k["name"], k["sgpr_spill_count"]) for k in kernels if k["sgpr_spill_count"] > limit(k["name"])]
    assert not bad, "kernels with SGPR spills: %s" % bad


def test_gemm_tiles_leave_room_for_two_workgroups_per_cu(kernels):
    """The 4-wave 128 x 128 GEMM tiles are meant to run two workgroups per CU (2 waves per SIMD): at most 256 of the
    512 registers per lane, arch + accumulator registers together."""
    seen = 0
    for k in kernels:
        if "gemm_kernel" in k["name"] and k["max_flat_workgroup_size"] == 256:
            seen += 1
            assert k["vgpr_count"] + k["agpr_count"] <= 256, (k["name"], k["vgpr_count"], k["agpr_count"])
    assert seen >= 4


def test_one_wave_variance_tiles_own_a_simd(kernels):
    """gpx_vargemm.hip: one 64-lane workgroup per SIMD with the whole accumulator file (256 AGPRs: 8 x 8 fp32 / 8 x 4 fp64
    fragments; 64 / 128 / 192 for the partial last row tiles with 2 / 4 / 6 row fragments) and at most 512 registers in
    all; four of them must fit a CU's 160 KiB of LDS (the fp64 epilogue of the fp32 tile stages its operands and one block
    of accumulators there)."""
    seen = 0
    for k in kernels:
        if "var_w1_" in k["name"]:
            seen += 1
            ni = 8
            for cand in (2, 4, 6):
                if "ELi%dE" % cand in k["name"]:
                    ni = cand
            assert k["max_flat_workgroup_size"] == 64, k["name"]
            assert k["agpr_count"] == 32 * ni, (k["name"], k["agpr_count"])
            assert k["vgpr_count"] <= 512, (k["name"], k["vgpr_count"])  # (the unified count: arch + accumulator registers)
            assert 4 * k["group_segment_fixed_size"] <= 160 * 1024, (k["name"], k["group_segment_fixed_size"])
    assert seen == 9  # fp32 {with the fit, plain} x {8, 6, 4, 2 row fragments}, fp64


def test_one_wave_main_loops_hold_nothing_but_mfmas_and_loads(gpx, tmp_path):
    """The main loops of gpx_vargemm.hip issue their MFMAs from inline asm, which hipcc's hazard recogniser and register
    allocator cannot see into: an accumulator copy (v_accvgpr_*) or a spill next to them would read a result that is
    still in flight (one build of the [k][n] kernel carried 64 such moves per trip until an early-return path was removed).
    Every backward branch of these kernels whose body holds >= 128 MFMAs must hold only MFMAs, buffer loads, waits and
    scalar / address arithmetic."""
    codeobj.guard_one_wave_main_loops(gpx.LIB_PATH, tmp_path)



def test_one_wave_tiles_never_touch_an_accumulator_between_their_mfmas(gpx, tmp_path):
    """Round 4 put the diagonal block of the one-wave tiles behind branches (zero fragments skipped).  At a merge hipcc may
    rename accumulators -- v_accvgpr_read / _mov / spills right behind asm MFMAs it cannot see into, i.e. reads of results
    still in flight (the first fp64 form did: 1e-6 errors at N = 4096).  From the first to the last accumulating MFMA of
    var_w1_kernel<with the fit, 8 fragments> and var_w1_f64_kernel nothing may read or move an accumulator register."""
    codeobj.guard_one_wave_accumulators(gpx.LIB_PATH, tmp_path)



def test_small_model_variance_kernels_run_two_waves_per_simd(kernels):
    """gpx_varcols_kernel.hpp: one 64-lane workgroup per wave, 12 x 2 accumulator fragments (96 AGPRs) and at most 256
    registers in all, so that two waves share a SIMD; eight workgroups must fit a CU's 160 KiB of LDS (the training points,
    the lane's queries and the column side of the fp64 add-back live there)."""
    seen = 0
    for k in kernels:
        if "var_cols_kernel" in k["name"]:
            seen += 1
            assert k["max_flat_workgroup_size"] == 64, k["name"]
            assert k["vgpr_count"] <= 256, (k["name"], k["vgpr_count"])  # (the unified count: arch + accumulator registers)
            assert k["agpr_count"] >= 96, (k["name"], k["agpr_count"])
            assert 8 * k["group_segment_fixed_size"] <= 160 * 1024, (k["name"], k["group_segment_fixed_size"])
    assert seen == 4  # operand formed in the wave for Gaussian / Laplace, Matern-3/2, Matern-5/2; operand read from the buffer


def test_small_model_variance_kernels_read_no_accumulator_in_flight(gpx, tmp_path):
    """The fp32 MFMAs of gpx_varcols_kernel.hpp are inline asm: hipcc neither sees their latency nor pads the hazards around
    them.  The epilogue of a row fragment reads its accumulators (v_accvgpr_read) from compiler-generated code placed behind
    later MFMAs; a read scheduled right behind the fragment's own last MFMA would fetch a result still in the pipe.  Every
    v_accvgpr_read of the kernels must therefore lie at least 4 MFMAs (128 cycles; the result is written after 8 passes = 32)
    or an explicit run of wait states behind the last MFMA that wrote the register."""
    codeobj.guard_small_model_accumulator_reads(gpx.LIB_PATH, tmp_path)



def test_small_fp64_variance_kernel_keeps_its_accumulators_in_place(gpx, tmp_path):
    """gpx_varcols64.hip: 14 accumulator slots tied to AGPRs and 8 to VGPRs through inline-asm MFMAs -- no accumulator copies
    (v_accvgpr_write / _mov) anywhere in the kernel, no scratch, and no read of an accumulator by anything but an MFMA before
    16 wait states or two later MFMAs have passed."""
    codeobj.guard_small_fp64_accumulators(gpx.LIB_PATH, tmp_path)


def test_small_split_fp16_variance_kernel_keeps_its_accumulators_in_place(gpx, tmp_path):
    """gpx_varcols16.hip: the same construction on v_mfma_f32_16x16x32_f16 (16 AGPR + 6 VGPR slots of 16 registers)."""
    codeobj.guard_small_split_accumulators(gpx.LIB_PATH, tmp_path)


def test_split_contraction_stages_by_lds_dma_behind_a_vmcnt_wait(gpx, tmp_path):
    """gpx_vsplit.hip: the k-tiles of the F32_SPLIT contraction arrive by LDS-DMA (global_load_lds_dwordx4), 8 per wave and
    tile, with no ds_write in the main loop; two workgroups must fit a CU (<= 256 registers, 64 KiB of LDS each).  An
    LDS-DMA write becomes visible to the other waves' ds_reads only through the issuing wave's `s_waitcnt vmcnt(0)` before
    the barrier -- hipcc left that wait out of one of the loop's two barriers until it was written into the source, so every
    s_barrier of the main loop must have one in the instructions in front of it."""
    codeobj.guard_split_contraction_staging(gpx.LIB_PATH, tmp_path)


def test_every_mfma_result_has_its_wait_states_on_every_path(gpx, tmp_path):
    """Round 6: the cause of the fp32 two-wave sub-block LDL^T's run-to-run differences (gpx_blk.hpp) -- hipcc's
    hazard recogniser sized the wait states behind an MFMA for the LONG way to the first read of its result (through the poll
    loop) and left the short way (poll skipped) 6-7 wait states where the ISA wants 18.  The guard walks every path of every
    MFMA kernel of the library."""
    codeobj.guard_mfma_result_hazards(gpx.LIB_PATH, tmp_path)


def _listing(rows):
    """fake llvm-objdump lines 'text // ADDR: ENC [<sym+0xOFF>]' from (text, branch-target index | None) rows"""
    out = []
    for i, (text, tgt) in enumerate(rows):
        tail = " <k+0x%x>" % (4 * tgt) if tgt is not None else ""
        out.append("%s // %012X: BF800000%s" % (text, 0x1000 + 4 * i, tail))
    return out


def test_hazard_walk_finds_the_short_path_the_compiler_missed():
    """The shape of the defect in miniature: behind a 16-pass MFMA a long path (a poll loop) and a short one (a taken branch)
    merge in front of the read of the accumulator; the s_nop in front of the read is sized for the long path.  A walk with a
    single visited set (long path first) calls this code safe; the guard's minimum over all paths must not."""
    rows = [
        ("v_mfma_f32_32x32x2_f32 a[0:15], v2, v3, a[0:15]", None),  # 0
        ("s_cbranch_scc1 10", 10),                                   # 1: short way -> merge
    ] + [("s_mov_b32 s0, s1", None)] * 8 + [                          # 2..9: the long way, 8 wait states
        ("s_nop 7", None),                                            # 10: merge block; 2 + 8 + 8 = 18 on the long path
        ("v_accvgpr_read_b32 v5, a3", None),                          # 11: read of the result
        ("s_endpgm", None),
    ]
    bad = codeobj.mfma_hazard_violations(_listing(rows))
    assert bad and bad[0][1] == "v_accvgpr_read_b32 v5, a3" and bad[0][2] == 9 and bad[0][3] == 18, bad
    # the same read behind enough wait states on BOTH paths, and an interlocked accumulate in between: clean
    rows[10] = ("s_nop 15", None)
    rows.insert(11, ("s_nop 1", None))
    assert not codeobj.mfma_hazard_violations(_listing(rows))
    rows2 = [("v_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]", None),
             ("v_mfma_f64_16x16x4_f64 a[0:7], v[2:3], v[4:5], a[0:7]", None)] + [("s_nop 4", None)] * 2 + [
             ("v_accvgpr_read_b32 v5, a3", None), ("s_endpgm", None)]
    bad = codeobj.mfma_hazard_violations(_listing(rows2))
    assert len(bad) == 1 and bad[0][2] == 10 and bad[0][3] == 11, bad  # (the second MFMA: 10 < 11; the first has 11)
