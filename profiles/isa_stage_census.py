#!/usr/bin/env python3
"""Static instruction census of the main-tier kernel by stage (build container, no GPU): device assembly with line tables
(hipcc -gline-tables-only --cuda-device-only -S), every instruction booked on the stage of the last vg_wave.h line seen before it.
Rough (inlined helpers and hoisted code land where the compiler put them; static counts, not executed ones -- lambdas inlined at
several call sites count several times), but it says how much code a pass through each stage is.
    python3 profiles/isa_stage_census.py            (writes /tmp/vg_dev.s)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "vargeno_amd", "csrc")
S = "/tmp/vg_dev.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "--cuda-device-only", "-S", "-o", S, os.path.join(C, "vargeno_hip.hip")], stderr=subprocess.DEVNULL)
lines = open(S).read().split("\n")
kern = "_ZN2vg14vg_wave_kernelILb0ELi14ELi6ELi4E"
start = next(i for i, l in enumerate(lines) if l.startswith(kern) and l.rstrip().endswith(":") or (l.startswith(kern) and ": ;" in l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
wave_file = next(int(re.match(r"\s*\.file\s+(\d+)", l).group(1)) for l in lines if re.match(r"\s*\.file\s+\d+.*vg_wave\.h", l))
src = open(os.path.join(C, "vg_wave.h")).read().split("\n")


def find(s):
    return next(i + 1 for i, l in enumerate(src) if s in l)


bounds = [(0, "prologue / refill"), (find("VG_CLK(0);"), "A: key_find / push_exact"), (find("auto push_row = [&]"), "A: push_row (auxiliary rows)"),
          (find("auto emit_exact = [&]"), "A: emit / no-merged-view / jump-table look-ups"), (find("if (d.dx) {"), "A: direct-table look-ups"), (find("VG_CLK(1);"), "B0"),
          (find("VG_CLK(2);"), "B1 set-up / scan_probe"), (find("for (uint32_t t0 = 0; t0 < T; t0 += 64) {"), "B1 round: decode"),
          (find("if (q_r || q_s) dual_query(d, hs, qk, q_r, q_s, ri, si);"), "B1 round: dual_query + acceptance"), (find("if (__any(keepm != 0)) {"), "B1 round: compaction"),
          (find("VG_CLK(3);"), "C: vote"), (find("VG_CLK(4);"), "C: walk"), (find("VG_CLK(5);"), "epilogue")]


def stage(ln):
    name = bounds[0][1]
    for b, n in bounds:
        if ln >= b:
            name = n
    return name


last, cnt, kinds = 0, collections.Counter(), collections.defaultdict(collections.Counter)
for i in range(start, end):
    l = lines[i].strip()
    if l.startswith(".loc"):
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
        if int(m.group(1)) == wave_file:
            last = int(m.group(2))
        continue
    if not l or l[0] in ".;" or l.endswith(":"):
        continue
    op = l.split()[0]
    if not re.match(r"^[sv]_|^buffer_|^global_|^flat_|^ds_|^scratch_", op):
        continue
    k = ("valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu" if op.startswith("s_")
         else "lds" if op.startswith("ds_") else "vmem")
    st = stage(last)
    cnt[st] += 1
    kinds[st][k] += 1
tot = sum(cnt.values())
print("static instructions of vg_wave_kernel<false, 14, 6, 4>: %d" % tot)
for _, n in bounds:
    if cnt[n]:
        print("%-48s %6d (%4.1f %%)  %s" % (n, cnt[n], 100.0 * cnt[n] / tot, "  ".join("%s %d" % kv for kv in sorted(kinds[n].items()))))
