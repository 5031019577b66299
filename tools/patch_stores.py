"""Experiment aid: set the sc1 (write-through) bit on the global stores of chosen kernels in a built libgscan_hip.so.

    python tools/patch_stores.py <in.so> <out.so> <kernel-name regex> [bits=sc1|sc0sc1|nt]

The code objects inside the library are uncompressed clang offload bundles; every gfx950 ELF is disassembled with
llvm-objdump, the `global_store_*` instructions of the kernels whose (mangled) name matches are located and bit 25
(sc1) / 16 (sc0) / 17 (nt) of their first dword is set in place.  The result is checked by disassembling it again.
Used to measure what written-through stores buy at kernel boundaries (tools/micro/store_policy_gap.hip) before the
stores that matter are changed in the sources.
"""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(data):
    i = 0
    while True:
        i = data.find(MAGIC, i)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts]
            p += ts
            if b"gfx950" in triple and size:
                yield i + off, size
        i += 1


def text_section(elf):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
    strtab = secs[shstrndx]
    for s in secs:
        name = elf[strtab[4] + s[0]:elf.index(b"\0", strtab[4] + s[0])]
        if name == b".text":
            return s[3], s[4]          # sh_addr, sh_offset
    raise SystemExit("no .text")


def store_sites(elf_bytes, pattern):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        dis = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout
    sites, cur, per = [], None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur and re.search(pattern, cur) and "global_store_" in line:
            m = re.search(r"// ([0-9A-Fa-f]+):", line)
            sites.append((int(m.group(1), 16), line.strip()))
            per[cur] = per.get(cur, 0) + 1
    return sites, per


def main():
    src, dst, pattern = sys.argv[1:4]
    bits = {"sc1": 1 << 25, "sc0sc1": (1 << 25) | (1 << 16), "nt": 1 << 17}[sys.argv[4] if len(sys.argv) > 4 else "sc1"]
    data = bytearray(open(src, "rb").read())
    total = 0
    for off, size in code_objects(bytes(data)):
        elf = bytes(data[off:off + size])
        addr, foff = text_section(elf)
        sites, per = store_sites(elf, pattern)
        for va, _ in sites:
            at = off + foff + (va - addr)
            w, = struct.unpack_from("<I", data, at)
            assert (w >> 26) == 0x37, hex(w)           # FLAT-family encoding
            struct.pack_into("<I", data, at, w | bits)
        for k, v in per.items():
            print(f"{v:5d} stores  {k[:110]}")
        total += len(sites)
        if sites:                                       # verify on the patched bytes
            again, _ = store_sites(bytes(data[off:off + size]), pattern)
            assert all("sc1" in l or "nt" in l for _, l in again), "patch did not take"
    open(dst, "wb").write(data)
    print(f"{total} stores patched -> {dst}")


if __name__ == "__main__":
    main()
