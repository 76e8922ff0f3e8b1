#!/usr/bin/env python
"""Identity of the device code of named kernels inside libloftr_hip.so (pure Python, no ROCm tools needed).

    python tools/kernel_code_hash.py [lib.so] [substring ...]

bench.py quotes PMC figures (profiles/pmc_traffic.json) only for the build they were collected on.  Until round 5 "the build" meant the
hash of every source file, so adding a training-only translation unit invalidated the forward kernels' counters although their machine code
had not changed by one byte.  The identity that matters for a counter is the code the GPU ran: every translation unit is its own gfx950 code
object inside the library's offload bundles; this module digs the kernels' function bodies out of those ELF images and hashes them
(sorted by symbol name), so a PMC table stays valid exactly as long as the kernels it describes are byte-identical."""
import hashlib
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """Yield the gfx950 ELF images of every offload bundle in the library."""
    pos = 0
    while True:
        b = blob.find(MAGIC, pos)
        if b < 0:
            return
        n = struct.unpack_from("<Q", blob, b + len(MAGIC))[0]
        p = b + len(MAGIC) + 8
        end = b + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode(errors="replace")
            p += 24 + tl
            if "gfx950" in triple and size:
                yield blob[b + off:b + off + size]
            end = max(end, b + off + size)
        pos = max(end, b + len(MAGIC))


def _functions(elf):
    """{symbol name: function bytes} of one ELF64 LE image."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    out = {}
    for (name, typ, flags, addr, off, size, link, info, align, entsize) in secs:
        if typ != 2:                       # SHT_SYMTAB
            continue
        strtab = secs[link]
        for i in range(size // entsize):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + i * entsize)
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= shnum:      # STT_FUNC
                continue
            e = elf.index(b"\0", strtab[4] + st_name)
            sym = elf[strtab[4] + st_name:e].decode()
            s = secs[st_shndx]
            body = elf[s[4] + (st_value - s[3]):s[4] + (st_value - s[3]) + st_size]
            out[sym] = body
    return out


def kernel_functions(lib_path):
    blob = open(lib_path, "rb").read()
    funcs = {}
    for elf in _code_objects(blob):
        funcs.update(_functions(elf))
    return funcs


def kernel_code_hash(lib_path, substrings):
    """16 hex digits over (name, body) of every device function whose mangled name contains one of `substrings`, sorted by name;
    (hash, number of functions)."""
    funcs = kernel_functions(lib_path)
    pick = sorted(n for n in funcs if any(s in n for s in substrings))
    h = hashlib.sha256()
    for n in pick:
        h.update(n.encode() + b"\0" + hashlib.sha256(funcs[n]).digest())
    return h.hexdigest()[:16], len(pick)


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else "loftr_amd/libloftr_hip.so"
    subs = sys.argv[2:]
    if subs:
        print(*kernel_code_hash(lib, subs))
    else:
        f = kernel_functions(lib)
        for n in sorted(f):
            print(hashlib.sha256(f[n]).hexdigest()[:16], len(f[n]), n[:140])
