#!/usr/bin/env python
"""What is inside a slow hipGraphLaunch?  (the one 7-8 ms replay early in a captured fit.)

    rocprofv3 --hip-trace --hsa-trace --output-format csv -d <dir> -- python bench.py ... --profile-run
    python tools/stall_trace.py <dir> [call name, default hipGraphLaunch] [how many of the slowest calls, default 3]

Reads the HIP and HSA API traces (and the memory-copy / kernel traces when present), ranks the calls of the named HIP
function by duration and, for each of the slowest, lists every HSA call that lies inside its time window on the same
thread -- grouped by function, with count and summed duration -- plus the calls' rank in launch order (is it the 3rd / 4th
replay?) and the median for comparison.  Prints plain text: meant to be committed under profiles/."""
import collections
import csv
import glob
import os
import sys


def rows(path):
    with open(path, newline="") as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            yield r


def col(r, *names):
    for n in names:
        for k in r:
            if k.lower() == n.lower():
                return r[k]
    for n in names:
        for k in r:
            if n.lower() in k.lower():
                return r[k]
    return None


def load(pattern, d):
    out = []
    for path in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        for r in rows(path):
            try:
                s, e = int(col(r, "Start_Timestamp", "start")), int(col(r, "End_Timestamp", "end"))
            except (TypeError, ValueError):
                continue
            out.append((s, e, col(r, "Function", "Name", "Kernel_Name") or "?", col(r, "Thread_Id", "tid") or "0"))
    out.sort()
    return out


def main():
    d = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "hipGraphLaunch"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    hip = load("*hip_api_trace.csv", d)
    hsa = load("*hsa_api_trace.csv", d)
    kern = load("*kernel_trace.csv", d)
    calls = [(e - s, i, s, e, t) for i, (s, e, f, t) in enumerate(c for c in hip if c[2] == name)]
    if not calls:
        names = collections.Counter(c[2] for c in hip).most_common(12)
        print(f"no call named {name}; most frequent HIP calls: {names}")
        return
    durs = sorted(c[0] for c in calls)
    med = durs[len(durs) // 2]
    print(f"{len(calls)} calls of {name}: median {med / 1e3:.1f} us, p99 {durs[int(0.99 * (len(durs) - 1))] / 1e3:.1f} us, "
          f"max {durs[-1] / 1e3:.1f} us; HIP rows {len(hip)}, HSA rows {len(hsa)}, kernel rows {len(kern)}")
    typical = sorted(calls, key=lambda c: abs(c[0] - med))[0]
    for label, pick in ([("TYPICAL (closest to the median)", typical)] +
                        [(f"SLOWEST #{k + 1}", c) for k, c in enumerate(sorted(calls, reverse=True)[:top])]):
        dur, idx, s, e, tid = pick
        inside = [h for h in hsa if h[0] >= s and h[1] <= e and h[3] == tid]
        other = [h for h in hsa if h[0] >= s and h[1] <= e and h[3] != tid]
        agg = collections.defaultdict(lambda: [0, 0])
        for hs, he, f, _ in inside:
            agg[f][0] += 1
            agg[f][1] += he - hs
        print(f"\n{label}: call #{idx} (in launch order), {dur / 1e3:.1f} us, thread {tid}; {len(inside)} HSA calls inside on "
              f"this thread ({sum(v[1] for v in agg.values()) / 1e3:.1f} us), {len(other)} on other threads")
        for f, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f"    {f:55s} x{n:5d}  {t / 1e3:10.1f} us")
        longest = sorted(inside, key=lambda h: h[0] - h[1])[:3]
        for hs, he, f, _ in longest:
            print(f"    longest single call: {f} {(he - hs) / 1e3:.1f} us at +{(hs - s) / 1e3:.1f} us")
        # device side: kernels that START inside the window, and the largest gap between consecutive kernel starts
        ks = [k for k in kern if s <= k[0] <= e]
        if len(ks) > 1:
            gap, at = max((ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1))
            print(f"    {len(ks)} kernels start inside; largest idle gap between them {gap / 1e3:.1f} us after {ks[at][2][:50]}")


if __name__ == "__main__":
    main()
