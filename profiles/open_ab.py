#!/usr/bin/env python3
"""Same-box A/B of vg_index_open's file stage: the ring of page-locked staging buffers the reader threads pread() the dictionary
files into (piece size x slots: $VG_FILE_PIECE_MB x $VG_FILE_RING).  Each variant opens the index in a process of its own (what a
job does), `idle` seconds after the one before (memory a process has just given back is cleared at the next allocation:
profiles/cold_start_r05.txt), and prints the open's wall time and its phase report.

    python3 profiles/open_ab.py <index prefix> <idle seconds> name:KEY=VALUE,KEY=VALUE ..."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from vargeno_amd.api import GenoIndex
t0 = time.time()
gx = GenoIndex.open(sys.argv[1], device=0)
dt = time.time() - t0
rep = gx.open_report
print(json.dumps({"open_s": round(dt, 3), "device_GB": round(gx.device_bytes / 1e9, 1), "sites": int(gx.num_sites), "report": rep}))
sys.stdout.flush()
os._exit(0)          # (no tear-down of 250 GB inside the measurement's process; the driver frees it)
''' % ROOT


def main():
    prefix, idle = sys.argv[1], float(sys.argv[2])
    for spec in sys.argv[3:]:
        name, _, kv = spec.partition(":")
        env = dict(os.environ)
        for item in [x for x in kv.split(",") if x]:
            k, _, v = item.partition("=")
            env[k] = v
        time.sleep(idle)
        p = subprocess.run([sys.executable, "-c", CHILD, prefix], env=env, capture_output=True, text=True, timeout=600)
        try:
            rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        except Exception:
            rec = {"failed": (p.stderr or "")[-400:]}
        rec.update({"variant": name, "settings": kv})
        files = [x for x in rec.get("report", "").split("; ") if x.startswith("block allocated")]
        rec["files_phase"] = files[0] if files else None
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
