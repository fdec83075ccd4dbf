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
