#!/usr/bin/env python3
"""How many GPU kernels does libmvdb.so hold, per translation unit, and does any of them spill?  Parses the clang offload
bundles inside the library (one per .hip file), extracts the gfx950 code objects and reads their symbol tables and
.AMDGPU metadata notes with llvm-readelf.  usage: kernel_census.py [path/to/libmvdb.so]  -> one JSON line."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "minivectordb_amd", "lib", "libmvdb.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
data = open(so, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
out = {"library": os.path.relpath(so, ROOT), "bytes": len(data), "code_objects": [], "kernels": 0, "spilling": []}
pos = 0
with tempfile.TemporaryDirectory() as tmp:
    while True:
        b = data.find(magic, pos)
        if b < 0:
            break
        n, = struct.unpack_from("<Q", data, b + len(magic))
        p = b + len(magic) + 8
        end = b
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            end = max(end, b + off + size)
            if "gfx950" not in triple or size == 0:
                continue
            path = os.path.join(tmp, f"co{len(out['code_objects'])}.elf")
            with open(path, "wb") as f:
                f.write(data[b + off:b + off + size])
            syms = subprocess.run([READELF, "-s", "--wide", path], capture_output=True, text=True).stdout
            kds = sorted({l.split()[-1][:-3] for l in syms.splitlines() if l.rstrip().endswith(".kd")})  # (.dynsym and .symtab list each)
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            spills = []
            # a kernel "spills" when it uses scratch memory or spills VGPRs (SGPRs parked in VGPR lanes cost no memory traffic)
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
                if int(m.group(2)) or int(m.group(3)):
                    spills.append({"kernel": m.group(1)[:80], "scratch_bytes": int(m.group(2)), "vgpr_spills": int(m.group(3))})
            fam = {}
            for k in kds:
                d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"<.*", "", d.split("(")[0]).split("::")[-1].replace("void ", "")
                fam[name] = fam.get(name, 0) + 1
            out["code_objects"].append({"bytes": size, "kernels": len(kds),
                                        "largest_families": dict(sorted(fam.items(), key=lambda kv: -kv[1])[:6])})
            out["kernels"] += len(kds)
            out["spilling"] += spills
        pos = max(end, b + len(magic))
print(json.dumps(out))
