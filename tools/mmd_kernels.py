#!/usr/bin/env python
"""Per-kernel durations of the MMD chain: tools/mmd_bench.py under rocprofv3 --kernel-trace --stats, averages read from the
result database (run on the GPU box).  python tools/mmd_kernels.py OUTDIR [reps]"""
import glob, json, os, sqlite3, subprocess, sys
out = sys.argv[1]
reps = sys.argv[2] if len(sys.argv) > 2 else "40"
here = os.path.dirname(os.path.abspath(__file__))
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "-o", "mmd", "--", sys.executable, os.path.join(here, "mmd_bench.py"), reps],
                   env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
rec = {"bench": ([l for l in r.stdout.splitlines() if l.startswith("loss")] or [r.stderr[-300:]])[-1]}
db = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
if db:
    cur = sqlite3.connect(db[0]).cursor()
    tot = 0.0
    for name, calls, avg in cur.execute("select name, total_calls, average from top_kernels"):
        if "k_" in name and "at::" not in name:
            rec[name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()[:40]] = round(avg, 2)
            tot += avg
    rec["sum_us"] = round(tot, 2)
print(json.dumps(rec))
