#!/usr/bin/env python
"""Condense rocprofv3 output (gpurun_out/, scratch) into the small tracked summaries under
profiles/: per-kernel time table from --kernel-trace --stats, and per-kernel HBM traffic from
the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: counters are in
KiB, and on gfx950 FETCH_SIZE reports half the bytes of a wide 16 B/lane coalesced read, so
the read side is doubled for kernels whose loads are 16 B/lane).

    python tools/summarize_rocprof.py --tag r1 --stats gpurun_out/prof_r1 \\
        --fetch gpurun_out/pmc_fetch_r1 --write gpurun_out/pmc_write_r1 [--bench gpurun_out/x.json]
"""
import argparse
import collections
import csv
import glob
import json
import os


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None       # newest run when a directory was reused


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--cmd", default=None, help="the profiled command, for the record")
    ap.add_argument("--lds", help="directory of a `--pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES ...` pass")
    ap.add_argument("--lds-kernel", default="k_kstep_lds", help="kernel name prefix the LDS section reports")
    ap.add_argument("--largest-grid", action="store_true",
                    help="PMC: per kernel keep only the dispatches with the largest grid (a sweep's biggest case)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines = [f"# rocprofv3 summary `{a.tag}`", ""]
    result = {"tag": a.tag}
    try:                  # the library the profiled command loaded: bench.py quotes this summary's durations beside its own
        import hashlib    # live ones and says so when the hashes differ
        lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pygda_amd", "libgda_hip.so")
        result["library_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    except Exception:
        result["library_sha16"] = None
    if a.bench and os.path.exists(a.bench):
        txt = [l for l in open(a.bench).read().splitlines() if l.startswith("{")]
        if txt:
            b = json.loads(txt[-1])
            result["bench"] = b
            cmd = a.cmd or f"python bench.py --steps {b['steps']} --warmup {b['warmup']} --no-cpu-baseline"
            lines += [f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}`", "",
                      f"bench line under the profiler: {b['ms_per_step']:.3f} ms/step, "
                      f"{b['value']:.4g} {b['unit']}", ""]
    if a.cmd and not (a.bench and os.path.exists(a.bench)):
        lines += [f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- {a.cmd}` "
                  "(PMC passes: the same command under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`)", ""]
    if a.stats:
        f = find(a.stats, "kernel_stats.csv")
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        lines += ["## Kernel time (`--kernel-trace --stats`)", "",
                  "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
        ks = []
        for r in rows[:40]:
            ks.append(dict(name=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                           avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"])))
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                         f"{float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                         f"{float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
        lines += ["", f"total kernel time {tot/1e6:.2f} ms over {len(rows)} distinct kernels", ""]
        result["kernels"] = ks
    pmc = {}
    for counter, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not d:
            continue
        f = find(d, "counter_collection.csv")
        agg = collections.defaultdict(list)
        rows_c = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        biggest = collections.defaultdict(int)
        for r in rows_c:
            biggest[short(r["Kernel_Name"])] = max(biggest[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        for r in rows_c:
            if a.largest_grid and int(r["Grid_Size"]) != biggest[short(r["Kernel_Name"])]:
                continue
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[counter] = dict(launches=len(v), avg_kib=sum(v) / len(v))
    if pmc:
        lines += ["## HBM-side traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)", "",
                  "FETCH_SIZE/WRITE_SIZE are KiB at the L2's fabric side (Infinity-Cache hits included). "
                  "`traffic` = 2 x FETCH + WRITE for kernels with 16 B/lane loads (gfx950 correction), "
                  "FETCH + WRITE otherwise." + (" Per kernel only the dispatches with the LARGEST grid are averaged "
                                                "(the sweep's biggest case)." if a.largest_grid else ""), "",
                  "| kernel | launches | FETCH KiB | WRITE KiB | traffic MB/launch |", "|---|---|---|---|---|"]
        ours = {}
        for k, v in sorted(pmc.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", {}).get("avg_kib", 0) *
                                                         kv[1].get("FETCH_SIZE", {}).get("launches", 0)))[:25]:
            fe, wr = v.get("FETCH_SIZE", {}).get("avg_kib", 0.0), v.get("WRITE_SIZE", {}).get("avg_kib", 0.0)
            wide = ((k.startswith("k_spmm<") or k.startswith("k_spmm_range<")) and k.split(",")[1].strip().startswith("4")) \
                or k.startswith("k_kstep_lds") or k.startswith("k_tall")
            traffic = ((2 if wide else 1) * fe + wr) * 1024
            n = v.get("FETCH_SIZE", v.get("WRITE_SIZE"))["launches"]
            lines.append(f"| `{k}` | {n} | {fe:.1f} | {wr:.1f} | {traffic/1e6:.3f} |")
            if k.startswith("k_"):
                ours[k] = dict(fetch_kib=fe, write_kib=wr, traffic_bytes=traffic, wide_correction=wide)
        result["pmc"] = ours
        lines.append("")
    if a.lds:
        f = find(a.lds, "counter_collection.csv")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(a.lds_kernel):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if agg:
            lines += ["## LDS counters per launch (separate `--pmc` pass; SQ counters are summed over the chip's SQs)", "",
                      "`SQ_LDS_IDX_ACTIVE` = LDS-array cycles spent on indexed (ds_read / ds_write) operations, "
                      "`SQ_LDS_BANK_CONFLICT` = the part of them that are bank-conflict replays "
                      "(MI355X_MICROARCH.md, LDS section): conflict fraction = BANK_CONFLICT / IDX_ACTIVE.", "",
                      "| kernel | launches | counter | avg per launch |", "|---|---|---|---|"]
            lds = {}
            for k, cs in agg.items():
                lds[k] = {}
                for c, v in sorted(cs.items()):
                    lines.append(f"| `{k}` | {len(v)} | {c} | {sum(v)/len(v):.4g} |")
                    lds[k][c] = dict(launches=len(v), avg=sum(v) / len(v))
                if "SQ_LDS_IDX_ACTIVE" in cs and "SQ_LDS_BANK_CONFLICT" in cs:
                    act, conf = sum(cs["SQ_LDS_IDX_ACTIVE"]), sum(cs["SQ_LDS_BANK_CONFLICT"])
                    lds[k]["bank_conflict_fraction"] = conf / act if act else None
                    lines.append(f"| `{k}` | | bank-conflict fraction of the LDS-array cycles | {conf / act if act else 0:.4f} |")
            result["lds"] = lds
            lines.append("")
    open(os.path.join(a.out, f"{a.tag}_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(result, open(os.path.join(a.out, f"{a.tag}_rocprof_summary.json"), "w"), indent=1)
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
