#!/usr/bin/env python
"""One of BASELINE.json's other configurations as bench.py's `other_configs` leg runs it (30 captured epochs, then a few
eager ones for the per-family kernel times) -- for profiling:

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/other_config.py grade_mmd|grade_js|udagcn|adagcn
    python tools/summarize_rocprof.py --tag r6_<name> --stats <dir> --cmd "python tools/other_config.py <name>" --out profiles
    python tools/step_timeline.py <dir> <k-th optimiser launch> <optimiser launches per epoch>     (one epoch's kernels)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                                   # noqa: E402

import bench                                                                   # noqa: E402


def main():
    which = sys.argv[1]
    r = bench.other_configs(torch.device("cuda:0"), which=(which,))
    print(which, r[which]["ms_per_epoch"], r[which].get("execution"))


if __name__ == "__main__":
    main()
