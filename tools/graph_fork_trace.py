"""Timeline of ONE replay of a forked graph (run under rocprofv3 --kernel-trace)."""
import sys
import torch
dev = "cuda:0"
mode = sys.argv[1]
n = 6
x = torch.randn(1 << 14, device=dev)
y = torch.randn(1 << 15, device=dev)      # chain B works on a different size: tell the chains apart by grid
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    r = x * 2.0
    ry = y * 2.0
    if mode == "side_first":
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b = ry
            for _ in range(n): b = b * 1.0001 + 0.5
        a = r
        for _ in range(n): a = a * 1.0001 + 0.5
        main.wait_stream(side)
    else:
        ev = torch.cuda.Event(); ev.record(main)
        a = r
        for _ in range(n): a = a * 1.0001 + 0.5
        side.wait_event(ev)
        with torch.cuda.stream(side):
            b = ry
            for _ in range(n): b = b * 1.0001 + 0.5
        main.wait_stream(side)
    out = a.sum() + b.sum()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
