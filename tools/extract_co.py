#!/usr/bin/env python3
"""Extract the gfx950 code object from a hipcc object file / shared library (clang offload bundle inside .hip_fatbin) and
disassemble it:  python tools/extract_co.py text_similarity_amd/csrc/_obj/k1_kl16.o /tmp/k1.s"""
import struct
import subprocess
import sys

data = open(sys.argv[1], "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = data.find(magic)
assert pos >= 0, "no offload bundle"
n = struct.unpack_from("<Q", data, pos + 24)[0]
o = pos + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", data, o)
    triple = data[o + 24:o + 24 + tl].decode()
    o += 24 + tl
    if "gfx950" in triple:
        co = sys.argv[2] + ".co"
        open(co, "wb").write(data[pos + off:pos + off + size])
        with open(sys.argv[2], "w") as f:
            subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], stdout=f, check=True)
        print(triple, size, "bytes ->", sys.argv[2])
