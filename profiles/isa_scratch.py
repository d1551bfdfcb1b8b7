#!/usr/bin/env python3
"""Development aid: where does a kernel touch scratch memory (register spills)?  Compiles vargeno_hip.hip with --save-temps
under /tmp/asm and lists the scratch_* instructions of the kernels whose mangled name contains the given substring, with
the nearest source line marker.   python3 profiles/isa_scratch.py ILb0ELi14 [-DKNOB=..]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/asm", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-g1", "--save-temps", *sys.argv[2:], "-c", "-o", "/tmp/asm/v.o",
                os.path.join(root, "vargeno_amd/csrc/vargeno_hip.hip")], cwd="/tmp/asm", stderr=subprocess.DEVNULL)
lines = open("/tmp/asm/vargeno_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
for start, l in enumerate(lines):
    if sys.argv[1] in l and l.startswith("_Z") and ":" in l and "; @" in l:
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        print(l.split(":")[0][:80], len(body), "lines")
        loc = ""
        n = 0
        for b in body:
            m = re.search(r"\.loc\s+\d+\s+(\d+)", b)
            if m:
                loc = m.group(1)
            if "scratch_" in b:
                n += 1
                print("   line", loc, b.strip()[:90])
        print("   scratch instructions:", n)
