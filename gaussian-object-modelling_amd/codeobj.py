"""Reads the AMDGPU code-object metadata (msgpack note NT_AMDGPU_METADATA) of every gfx950 kernel inside a host
shared library built by hipcc: the .hip_fatbin section is a sequence of clang offload bundles, each holding one
device ELF per target.  Used by the CPU-side checks on register spills / scratch (tests/test_codeobj.py) and by
scripts/kernel_resources.py."""
import struct
import subprocess

import msgpack

OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(path, tmp):
    out = str(tmp) + "/fat.bin"
    subprocess.run([OBJCOPY, "--dump-section", ".hip_fatbin=" + out, path], check=True)
    return open(out, "rb").read()


def _device_elfs(fat):
    pos = fat.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", fat, pos + 24)[0]
        off = pos + 32
        for _ in range(n):
            eo, es, tl = struct.unpack_from("<QQQ", fat, off)
            triple = fat[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and es > 0:
                yield fat[pos + eo:pos + eo + es]
        pos = fat.find(MAGIC, pos + 24)


def _notes(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = elf[shoff + i * shentsize:shoff + (i + 1) * shentsize]
        stype = struct.unpack_from("<I", sh, 4)[0]
        if stype != 7:  # SHT_NOTE
            continue
        o, sz = struct.unpack_from("<QQ", sh, 0x18)
        p, end = o, o + sz
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                yield msgpack.unpackb(desc, raw=False, strict_map_key=False)


def kernels(lib_path, tmp_dir):
    """[{name, vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size,
    group_segment_fixed_size, max_flat_workgroup_size}, ...] for every gfx950 kernel in lib_path."""
    out = []
    for elf in _device_elfs(_fatbin(lib_path, tmp_dir)):
        for md in _notes(elf):
            for k in md.get("amdhsa.kernels", []):
                out.append({"name": k[".name"], "vgpr_count": k.get(".vgpr_count", 0), "agpr_count": k.get(".agpr_count", 0),
                            "sgpr_count": k.get(".sgpr_count", 0), "vgpr_spill_count": k.get(".vgpr_spill_count", 0),
                            "sgpr_spill_count": k.get(".sgpr_spill_count", 0),
                            "private_segment_fixed_size": k.get(".private_segment_fixed_size", 0),
                            "group_segment_fixed_size": k.get(".group_segment_fixed_size", 0),
                            "max_flat_workgroup_size": k.get(".max_flat_workgroup_size", 0)})
    return out


def disassemble(lib_path, tmp_dir, name_part):
    """{kernel symbol: [instruction text, ...]} for the gfx950 kernels of lib_path whose symbol contains name_part
    (llvm-objdump -d of the device ELFs; branch targets keep their '<symbol+0xoffset>' form)."""
    out = {}
    for n, elf in enumerate(_device_elfs(_fatbin(lib_path, tmp_dir))):
        path = "%s/dev%d.elf" % (tmp_dir, n)
        open(path, "wb").write(elf)
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            if line.endswith(">:") and "<" in line:
                sym = line[line.index("<") + 1:-2]
                cur = sym if name_part in sym else None
                if cur:
                    out[cur] = []
            elif cur and line.strip():
                out[cur].append(line.strip())
    return out


# ---- guards on the kernels whose MFMAs are inline asm -----------------------------------------------------------------
# hipcc neither sees the latency of an asm MFMA nor pads the hazards around it, so the correctness of these kernels rests
# on what the compiler put BETWEEN them -- checked here on the disassembly of the built library.  __graft_entry__.build()
# runs check_library() on every build (a toolchain that reorders around the asm MFMAs fails the build, not a later test);
# tests/test_codeobj.py runs the same functions one by one.
def guard_one_wave_main_loops(lib_path, tmp_dir):
    """The main loops of gpx_vargemm.hip issue their MFMAs from inline asm, which hipcc's hazard recogniser and register
    allocator cannot see into: an accumulator copy (v_accvgpr_*) or a spill next to them would read a result that is
    still in flight (one build of the [k][n] kernel carried 64 such moves per trip until an early-return path was removed).
    Every backward branch of these kernels whose body holds >= 128 MFMAs must hold only MFMAs, buffer loads, waits and
    scalar / address arithmetic."""
    dis = disassemble(lib_path, tmp_dir, "w1_")
    assert len(dis) == 10, sorted(dis)  # var_w1_kernel<true|false, 8|6|4|2>, var_w1_f64_kernel, w1_f64_nn_kernel
    for sym, lines in dis.items():
        # 'mnemonic operands   // ADDRESS: ENCODING [<symbol+0xOFFSET>]'
        ins = [(int(l.split("//")[1].split(":")[0], 16), l.split("//")[0].strip(), l) for l in lines if "//" in l]
        first = ins[0][0]
        spans = []  # backward branches whose body holds the MFMAs; the innermost one is the main loop (the paired launch wraps it)
        for a, text, raw in ins:
            if not text.startswith("s_cbranch") or "+0x" not in raw:
                continue
            target = first + int(raw[raw.rindex("+0x") + 3:raw.rindex(">")], 16)
            if target < a and sum(t.startswith("v_mfma") for b, t, _ in ins if target <= b <= a) >= 128:
                spans.append((a - target, target, a))
        assert spans, sym
        _, lo, hi = min(spans)
        body = [t for b, t, _ in ins if lo <= b <= hi]
        bad = [t for t in body if t.startswith(("v_accvgpr", "scratch_", "v_mov_b", "ds_"))]
        assert not bad, (sym, bad[:5])


def guard_one_wave_accumulators(lib_path, tmp_dir):
    """Round 4 put the diagonal block of the one-wave tiles behind branches (zero fragments skipped).  At a merge hipcc may
    rename accumulators -- v_accvgpr_read / _mov / spills right behind asm MFMAs it cannot see into, i.e. reads of results
    still in flight (the first fp64 form did: 1e-6 errors at N = 4096).  From the first to the last accumulating MFMA of
    var_w1_kernel<with the fit, 8 fragments> and var_w1_f64_kernel nothing may read or move an accumulator register."""
    dis = disassemble(lib_path, tmp_dir, "var_w1_")
    seen = 0
    for sym, lines in dis.items():
        if "var_w1_f64_kernel" in sym:
            mf, expect = "v_mfma_f64_16x16x4_f64 a", 128 + 2 * 576   # main loop + the diagonal pairs, ascending and descending
        elif "var_w1_kernelILb1ELi8" in sym:
            mf, expect = "v_mfma_f32_16x16x4_f32 a", 512 + 2 * 512   # main loop + one rolled pair of diagonal chunks per direction
        else:
            continue
        seen += 1
        text = [l.split("//")[0].strip() for l in lines]
        idx = [i for i, t in enumerate(text) if t.startswith(mf)]
        assert len(idx) == expect, (sym, len(idx))
        bad = [t for t in text[idx[0]:idx[-1]] if t.startswith(("v_accvgpr_read", "v_accvgpr_mov", "scratch_"))]
        assert not bad, (sym, bad[:5])
    assert seen == 2


def guard_small_model_accumulator_reads(lib_path, tmp_dir):
    """The fp32 MFMAs of gpx_varcols_kernel.hpp are inline asm: hipcc neither sees their latency nor pads the hazards around
    them.  The epilogue of a row fragment reads its accumulators (v_accvgpr_read) from compiler-generated code placed behind
    later MFMAs; a read scheduled right behind the fragment's own last MFMA would fetch a result still in the pipe.  Every
    v_accvgpr_read of the kernels must therefore lie at least 4 MFMAs (128 cycles; the result is written after 8 passes = 32)
    or an explicit run of wait states behind the last MFMA that wrote the register."""
    import re
    dis = disassemble(lib_path, tmp_dir, "var_cols_kernel")
    assert len(dis) == 4, sorted(dis)
    for sym, lines in dis.items():
        last_write = {}  # accumulator register -> index (in MFMAs) of the last MFMA that wrote it
        nops_since = {}  # accumulator register -> wait states (s_nop) seen since that MFMA
        n_mfma = reads = 0
        for l in lines:
            text = l.split("//")[0].strip()
            if text.startswith("v_mfma_f32"):
                m = re.match(r"v_mfma_f32\S*\s+a\[(\d+):(\d+)\]", text)
                assert m, text
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    last_write[r] = n_mfma
                    nops_since[r] = 0
                n_mfma += 1
            elif text.startswith("s_nop"):
                w = int(text.split()[1]) + 1
                for r in nops_since:
                    nops_since[r] += w
            elif text.startswith("v_accvgpr_read"):
                r = int(re.search(r"\ba(\d+)\b", text).group(1))
                if r in last_write:  # (registers the compiler parks values in are never MFMA destinations)
                    reads += 1
                    assert n_mfma - last_write[r] >= 4 or nops_since[r] >= 16, (sym, text, n_mfma - last_write[r], nops_since[r])
        assert n_mfma >= 500 and reads >= 96, (sym, n_mfma, reads)
    # The fp64 MFMAs of the add-back (asm as well) write VGPR quads that plain VALU code reads.  The matrix pipe completes in
    # order, so ONE later MFMA issued between the last v_mfma_f64 that wrote a register and its first VALU read means the fp64
    # result has left the pipe (the later MFMA could not start before it); an explicit run of >= 16 wait states serves too.
    # (the source comment promised "at least NM" MFMAs in between, the 6-group chunk has one.)
    def vregs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return range(int(m.group(1)), int(m.group(2)) + 1)
        m = re.fullmatch(r"v(\d+)", tok)
        return range(int(m.group(1)), int(m.group(1)) + 1) if m else range(0)
    for sym, lines in dis.items():
        last64 = {}  # VGPR -> (index in MFMAs of the last v_mfma_f64 that wrote it, wait states since)
        n_mfma = checked = 0
        for l in lines:
            text = l.split("//")[0].strip()
            if not text:
                continue
            op, _, rest = text.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            if op.startswith("v_mfma"):
                n_mfma += 1
                if op.startswith("v_mfma_f64"):
                    for r in vregs(toks[0]):
                        last64[r] = [n_mfma, 0]
                continue
            if op == "s_nop":
                for v in last64.values():
                    v[1] += int(toks[0]) + 1
                continue
            if not op.startswith("v_") or not last64:
                continue
            for t in toks[1:]:  # sources
                for r in vregs(t):
                    if r in last64:
                        checked += 1
                        w, nops = last64.pop(r)  # (only the FIRST read after the write is the hazard)
                        assert n_mfma - w >= 1 or nops >= 16, (sym, text, n_mfma - w, nops)
            for r in vregs(toks[0]):  # overwritten by plain VALU code: no longer an MFMA result
                last64.pop(r, None)
        assert checked >= 32, (sym, checked)


def guard_small_fp64_accumulators(lib_path, tmp_dir):
    """gpx_varcols64.hip: 22 x 2 fp64 accumulator fragments per wave, 14 slots tied to AGPRs and 8 to VGPRs through inline-asm
    MFMAs.  (1) Every definition of an AGPR accumulator is an MFMA (zeroing included): a v_accvgpr_write / _mov in the kernel
    means the register allocator gave accumulators a second home again and copies them around the MFMA statements (one
    build carried 430 such moves in the chunk loop).  (2) No scratch.  (3) hipcc pads no hazard behind an asm MFMA: the first
    read of an accumulator by anything but an MFMA (v_accvgpr_read, or a VALU source for the VGPR slots) must lie behind
    >= 16 wait states or two later MFMAs (the matrix pipe completes in order)."""
    _guard_asm_accumulators(lib_path, tmp_dir, "VC64One", 4, "v_mfma_f64", 16 * 8 + 28, 14 * 16)
    # round 6, two waves per SIMD: 22 x 1 fragments per wave, 16 slots in AGPRs and 6 in VGPRs -- the one split that fits the 128 +
    # 128 registers hipcc gives a kernel of 512 threads; every other split measured made it copy accumulators right behind the asm
    # MFMAs (wrong variances): conditions (1) and (3) are what fail such a build
    _guard_asm_accumulators(lib_path, tmp_dir, "VC64Two", 4, "v_mfma_f64", 16 + 22 * 4, 16 * 8)
    for k in kernels(lib_path, tmp_dir):
        if "var_cols64_kernel" in k["name"] and "VC64Two" in k["name"]:
            assert k["vgpr_count"] <= 256 and k["private_segment_fixed_size"] == 0, k


def guard_small_split_accumulators(lib_path, tmp_dir):
    """gpx_varcols16.hip, the same construction on v_mfma_f32_16x16x32_f16: 22 slots of a main and a correction accumulator for
    two column fragments (16 registers per slot), 16 slots tied to AGPRs and 6 to VGPRs.  Same three conditions (the fp64
    MFMAs of its add-back are compiler builtins, whose hazards hipcc pads itself: only the asm ones are tracked)."""
    _guard_asm_accumulators(lib_path, tmp_dir, "var_cols16_kernel", 3, "v_mfma_f32_16x16x32", 22 * 6 + 64, 16 * 16)


def _guard_asm_accumulators(lib_path, tmp_dir, name, n_kernels, mfma_op, min_mfma, min_reads):
    import re
    dis = disassemble(lib_path, tmp_dir, name)
    assert len(dis) == n_kernels, sorted(dis)

    def regs(tok, cls):
        m = re.fullmatch(cls + r"\[(\d+):(\d+)\]", tok)
        if m:
            return range(int(m.group(1)), int(m.group(2)) + 1)
        m = re.fullmatch(cls + r"(\d+)", tok)
        return range(int(m.group(1)), int(m.group(1)) + 1) if m else range(0)
    for sym, lines in dis.items():
        pending = {}  # ('a' | 'v', register) -> [MFMAs issued when it was last written by an MFMA, wait states since]
        n_mfma = reads = 0
        for l in lines:
            text = l.split("//")[0].strip()
            if not text:
                continue
            op, _, rest = text.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            assert not op.startswith("scratch_"), (sym, text)
            assert op not in ("v_accvgpr_write_b32", "v_accvgpr_mov_b32"), (sym, text)
            if op.startswith("v_mfma"):
                n_mfma += 1
                if op.startswith(mfma_op):
                    for cls in "av":
                        for r in regs(toks[0], cls):
                            pending[(cls, r)] = [n_mfma, 0]
                continue
            if op == "s_nop":
                for v in pending.values():
                    v[1] += int(toks[0]) + 1
                continue
            if op == "v_accvgpr_read_b32":
                srcs = [("a", r) for r in regs(toks[1], "a")]
            elif op.startswith("v_"):
                srcs = [("v", r) for t in toks[1:] for r in regs(t, "v")]
            else:
                srcs = []
            for key in srcs:
                if key in pending:
                    reads += 1
                    w, nops = pending.pop(key)  # (only the first read behind the write is the hazard)
                    assert n_mfma - w >= 2 or nops >= 16, (sym, text, n_mfma - w, nops)
            if op.startswith("v_") or op.startswith("global_load") or op.startswith("ds_read"):
                for r in regs(toks[0], "v"):  # overwritten by other code: no longer an MFMA result
                    pending.pop(("v", r), None)
        assert n_mfma >= min_mfma and reads >= min_reads, (sym, n_mfma, reads)


def guard_split_contraction_staging(lib_path, tmp_dir):
    """gpx_vsplit.hip: the k-tiles of the F32_SPLIT contraction arrive by LDS-DMA (global_load_lds_dwordx4), 8 per wave and
    tile, with no ds_write in the main loop; two workgroups must fit a CU (<= 256 registers, 64 KiB of LDS each).  An
    LDS-DMA write becomes visible to the other waves' ds_reads only through the issuing wave's `s_waitcnt vmcnt(0)` before
    the barrier -- hipcc left that wait out of one of the loop's two barriers until it was written into the source, so every
    s_barrier of the main loop must have one in the instructions in front of it."""
    ks = [k for k in kernels(lib_path, tmp_dir) if "vsplit_gemm_kernel" in k["name"]]
    assert len(ks) == 1
    assert ks[0]["vgpr_count"] + ks[0]["agpr_count"] <= 256, ks[0]
    dis = disassemble(lib_path, tmp_dir, "vsplit_gemm_kernel")
    assert len(dis) == 1, sorted(dis)
    lines = next(iter(dis.values()))
    ins = [(int(l.split("//")[1].split(":")[0], 16), l.split("//")[0].strip(), l) for l in lines if "//" in l]
    first = ins[0][0]
    spans = []
    for a, text, raw in ins:
        if text.startswith("s_cbranch") and "+0x" in raw:
            target = first + int(raw[raw.rindex("+0x") + 3:raw.rindex(">")], 16)
            if target < a and sum(t.startswith("v_mfma") for b, t, _ in ins if target <= b <= a) >= 96:
                spans.append((a - target, target, a))
    assert spans
    _, lo, hi = min(spans)
    body = [t for b, t, _ in ins if lo <= b <= hi]
    assert sum(t.startswith("v_mfma_f32_16x16x32_f16") for t in body) == 96   # 2 k-tiles x 16 fragment pairs x 3 products
    assert sum(t.startswith("global_load_lds_dwordx4") for t in body) == 16  # 2 k-tiles x 2 operands x 4 pieces per wave
    assert sum(t.startswith("ds_read_b128") for t in body) == 32
    assert not [t for t in body if t.startswith(("ds_write", "scratch_", "global_load_dword", "buffer_load"))]
    barriers = [i for i, t in enumerate(body) if t.startswith("s_barrier")]
    assert len(barriers) == 2
    for i in barriers:
        assert any("vmcnt(0)" in t for t in body[max(0, i - 3):i]), body[max(0, i - 3):i + 1]



# ---- MFMA result hazards across the control-flow graph ------------------------------------------------------------------
# gfx950 does not interlock a read of an MFMA result that is still in the pipe: the ISA asks for software wait states between
# the MFMA and the first VALU / LDS / memory instruction that reads (or overwrites) its destination.  hipcc's hazard recogniser
# pads them -- but it walks the CFG backwards with ONE visited set: a block first reached through a long path is not walked
# again through a shorter one, so at a merge of a long path (a poll loop) and a short one (the poll skipped) the padding is
# sized for the long path.  That was the "unexplained race" of the fp32 two-wave sub-block LDL^T (gpx_blk.hpp):
# v_accvgpr_read of the inverse's accumulator 6-7 wait states behind a 16-pass v_mfma_f32_32x32x2_f32 on the path that needs
# no poll, i.e. only when wave 0 happens to be two columns ahead -- results that differ from run to run.  This walk takes the
# true minimum over all paths.
def _mfma_wait_states(op):
    """(wait states before a VALU read/write of the result, before an LDS / memory read) for the MFMA forms this library uses:
    LLVM's GCNHazardRecognizer tables for gfx940 / gfx950 (SGEMM = f32 inputs: passes + 2; XDL = f16 / bf16 / i8 / f8: passes + 3,
    one more for the 4-pass forms on gfx950; DGEMM 16x16x4: 11 / 18, 4x4x4: 6 / 9)."""
    if op.startswith("v_mfma_f64_16x16x4"):
        return 11, 18
    if op.startswith("v_mfma_f64_4x4x4"):
        return 6, 9
    shape = op.split("_")[3] if op.count("_") >= 3 else ""
    if op.startswith("v_mfma_f32") and op.endswith("_f32") and not op.endswith("xf32"):
        passes = {"32x32x2": 16, "32x32x1": 16, "16x16x4": 8, "16x16x1": 8, "4x4x1": 2}.get(shape)
        assert passes, op
        return passes + 2, passes + 2
    passes = {"32x32x16": 8, "16x16x32": 4, "32x32x8": 8, "16x16x16": 4, "32x32x4": 16, "16x16x8": 8, "4x4x4": 2,
              "32x32x64": 16, "16x16x128": 8, "32x32x32": 8, "16x16x64": 4}.get(shape, 16)
    w = passes + 3 + (1 if passes == 4 else 0)
    return w, w


def _reg_set(tok):
    import re
    m = re.fullmatch(r"-?\|?([av])\[(\d+):(\d+)\]\|?", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"-?\|?([av])(\d+)\|?", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def mfma_hazard_violations(lines):
    """lines: the disassembly of ONE kernel (codeobj.disassemble).  Returns [(mfma text, offending text, wait states found,
    wait states required)] for every instruction that touches the destination of an MFMA on SOME path with fewer wait states
    than the ISA requires (another MFMA accumulating into the same registers is interlocked and not counted)."""
    import heapq
    ins = [(int(l.split("//")[1].split(":")[0], 16), l.split("//")[0].strip(), l) for l in lines if "//" in l]
    if not ins:
        return []
    first = ins[0][0]
    at = {a: i for i, (a, _, _) in enumerate(ins)}
    n = len(ins)
    ops, toks, succ, ws = [], [], [], []
    for i, (a, text, raw) in enumerate(ins):
        op, _, rest = text.partition(" ")
        ops.append(op)
        toks.append([t.strip() for t in rest.split(",")] if rest else [])
        nxt = [i + 1] if i + 1 < n else []
        if op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
            nxt = []
        elif op == "s_branch" or op.startswith("s_cbranch"):
            tgt = []
            if "+0x" in raw:
                t = first + int(raw[raw.rindex("+0x") + 3:raw.rindex(">")], 16)
                if t in at:
                    tgt = [at[t]]
            elif raw.rstrip().endswith(">") and "<" in raw:  # '<symbol>' = offset 0
                tgt = [0]
            nxt = tgt if op == "s_branch" else tgt + nxt
        succ.append(nxt)
        ws.append(int(toks[-1][0]) + 1 if op == "s_nop" else 1)
    stores = ("ds_write", "ds_store", "global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic", "ds_add",
              "ds_max", "ds_min", "buffer_atomic", "flat_atomic")
    out = []
    for i in range(n):
        if not ops[i].startswith("v_mfma") and not ops[i].startswith("v_smfmac"):
            continue
        dst = _reg_set(toks[i][0])
        need_valu, need_mem = _mfma_wait_states(ops[i])
        limit = max(need_valu, need_mem)
        best = {}
        heap = [(0, j) for j in succ[i]]
        while heap:
            w, j = heapq.heappop(heap)
            if w >= limit or best.get(j, limit) <= w:
                continue
            best[j] = w
            op = ops[j]
            if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                # srcC / vdst on the same registers: the matrix pipe orders them itself; A / B operands are plain reads
                touched = set().union(*[_reg_set(t) for t in toks[j][1:3]]) if len(toks[j]) >= 3 else set()
                need = need_valu + 1
            else:
                src = toks[j] if op.startswith(stores) else toks[j][1:]
                touched = set().union(*[_reg_set(t) for t in src]) if src else set()
                need = need_valu if op.startswith("v_") else need_mem
                if toks[j] and not op.startswith(stores):
                    over = _reg_set(toks[j][0])  # overwriting a result in flight is the same hazard ...
                    if op.startswith("v_"):
                        touched |= over
                    elif over & dst and not (touched & dst):
                        # ... for a VALU write.  A LOAD that only overwrites (dead) registers of the destination writes them when its
                        # data returns, at least an LDS / L1 latency (> 16 wait states) after its issue: LLVM's recogniser does not
                        # count it at all; here it must still sit behind the wait states of a VALU read (round 6: the rank-4 LDL^T
                        # reads element 0 of a transform MFMA's tile, a ds_read2_b64 reused the tile's last registers 15 wait states
                        # behind a DGEMM that needs 11 / 18)
                        touched |= over
                        need = need_valu
            if w < need and touched & dst:
                out.append((ins[i][1], ins[j][1], w, need))
                continue
            if touched & dst and not (op.startswith("v_mfma") or op.startswith("v_smfmac")):
                # (past the first legal touch of these registers the MFMA has retired as far as they are concerned; keep
                # walking for the other registers of the destination)
                pass
            for k in succ[j]:
                heapq.heappush(heap, (w + ws[j], k))
    return out


def guard_mfma_result_hazards(lib_path, tmp_dir):
    """EVERY kernel of the library that issues an MFMA must have the ISA's wait states on every path from the MFMA to the first
    touch of its result (56 kernels in the round-6 build; the two-wave sub-block LDL^T of the dataflow factorisation is the one
    that failed: 26 short paths per kernel before its wave-1 MFMAs carried their own wait states)."""
    seen = pairs = 0
    for sym, lines in disassemble(lib_path, tmp_dir, "").items():
        if not any("v_mfma" in l or "v_smfmac" in l for l in lines):
            continue
        seen += 1
        bad = mfma_hazard_violations(lines)
        assert not bad, (sym, bad[:6], len(bad))
        if "factor_kernel" in sym and "wide" not in sym:
            # the hand-over protocol itself: wave 1's updates are asm MFMAs followed by their wait states in the same statement
            text = [l.split("//")[0].strip() for l in lines]
            idx = [i for i, t in enumerate(text) if t.startswith(("v_mfma_f64_16x16x4_f64", "v_mfma_f32_32x32x2_f32"))
                   and i + 2 < len(text) and text[i + 1] == "s_nop 15" and text[i + 2] == "s_nop 1"]
            # (fp32: one per column and sub-block = 2 x 32; fp64, rank-4 panels: 8 + 4 transforms and 4 + 6 updates per sub-block = 2 x 22)
            assert len(idx) >= 2 * 22, (sym, len(idx))
            pairs += 1
    assert seen >= 50 and pairs >= 12, (seen, pairs)


def check_library(lib_path):
    """Every guard above on lib_path; raises AssertionError naming the kernel and the offending instructions."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        guard_one_wave_main_loops(lib_path, td)
        guard_one_wave_accumulators(lib_path, td)
        guard_small_model_accumulator_reads(lib_path, td)
        guard_small_fp64_accumulators(lib_path, td)
        guard_small_split_accumulators(lib_path, td)
        guard_split_contraction_staging(lib_path, td)
        guard_mfma_result_hazards(lib_path, td)
