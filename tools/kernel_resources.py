#!/usr/bin/env python3
"""Registers, spills and scratch of the kernels in the SHIPPED library, from the code objects embedded in it
(llvm-readelf --notes on every gfx950 ELF of the .hip_fatbin): what DESIGN.md 3.1 quotes.

    python tools/kernel_resources.py [sigtk_amd/libsigtk_gpu.so] [name filter ...]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    i = 0
    while True:
        j = data.find(b"\x7fELF", i)
        if j < 0:
            return
        if data[j + 4] == 2 and data[j + 18:j + 20] == b"\xe0\x00":   # 64-bit, EM_AMDGPU
            e_shoff = struct.unpack_from("<Q", data, j + 0x28)[0]
            e_shentsize, e_shnum = struct.unpack_from("<HH", data, j + 0x3A)
            size = e_shoff + e_shentsize * e_shnum
            yield data[j:j + size]
            i = j + size
        else:
            i = j + 4


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    path = args.pop(0) if args and os.path.exists(args[0]) else os.path.join(root, "sigtk_amd", "libsigtk_gpu.so")
    rows = set()
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if args and not any(a in name for a in args):
                continue
            rows.add("%-72s vgpr %3s  vgpr_spill %3s  sgpr_spill %3s  scratch %4s B  lds %6s B" %
                     (name[:72], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                      g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    print("\n".join(sorted(rows)))


if __name__ == "__main__":
    main()
