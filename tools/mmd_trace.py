#!/usr/bin/env python
"""Where does a workgroup of the chunked one-pass MMD spend its time?  Needs a tracing build of gda_mmd.hip:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGDA_MMD_TRACE -shared pygda_amd/csrc/gda_mmd.hip -o <lib.so>
    python tools/mmd_trace.py <lib.so> [d, default 128] [resamples, default 5; 1 = at most one workgroup per CU]

Calls gda_mmd_chunked_fwd_f32 at (times 5, 1000 rows per domain, d) a few times and prints, from the per-workgroup clock
stamps of the last call (s_memtime at the phase boundaries, thread 0 of every workgroup): mean / max cycles of prologue,
phase 1 (distances), phase 2 (exponentials), phase 3 (gradient), tail, and when workgroups started and ended relative to
the first start (are there two rounds?)."""
import ctypes
import sys

import torch

P = ctypes.c_void_p


def main():
    lib = ctypes.CDLL(sys.argv[1])
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    times, n = (int(sys.argv[3]) if len(sys.argv) > 3 else 5), 1000
    dev = torch.device("cuda:0")
    plan = (ctypes.c_int64 * 8)()
    assert lib.gda_mmd_chunked_plan(times, ctypes.c_int64(n), ctypes.c_int64(d), ctypes.c_float(2.0), 5, plan, 8) == 0
    nseg, dp, nb, nc, ntiles, njb, total, img = list(plan)
    print(f"d {d}: nb {nb} nc {nc} dp {dp} ntiles {ntiles} njb {njb} nseg {nseg} workgroups {total}")
    lib.gda_mmd_chunked_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib.gda_mmd_chunked_workspace_bytes(times, ctypes.c_int64(n), ctypes.c_int64(d))
    gen = torch.Generator().manual_seed(1)
    s = torch.randn(times * n, d, generator=gen).relu().to(dev)
    t = (torch.randn(times * n, d, generator=gen) * 1.3 + 0.2).relu().to(dev)
    rows_s, rows_t = torch.empty(times * n, dp, device=dev), torch.empty(times * n, dp, device=dev)
    part = torch.empty(times, nseg, 2 * n, dp, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    loss, bw = torch.empty(1, device=dev), torch.empty(times, device=dev)
    trace = torch.zeros((total + 8) * 8, dtype=torch.int64, device=dev)
    assert lib.gda_dbg_mmd_trace(P(trace.data_ptr())) == 0
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        st = lib.gda_mmd_chunked_fwd_f32(
            P(s.data_ptr()), ctypes.c_int64(d), P(t.data_ptr()), ctypes.c_int64(d), ctypes.c_int64(d), None, None, times,
            ctypes.c_int64(n), ctypes.c_float(2.0), 5, ctypes.c_float(0.0), ctypes.c_float(1.0), None, P(rows_s.data_ptr()),
            P(rows_t.data_ptr()), ctypes.c_int64(dp), P(loss.data_ptr()), P(bw.data_ptr()), P(part.data_ptr()),
            ctypes.c_int64(dp), int(nseg), P(ws.data_ptr()), ctypes.c_size_t(wsb), P(stream))
        assert st == 0, st
    torch.cuda.synchronize()
    tr = trace.view(-1, 8)[:total].cpu().double()
    t0 = tr[:, 0].min()
    names = ["prologue", "phase 1", "phase 2", "phase 3", "tail"]
    for i, nm in enumerate(names):
        dt = tr[:, i + 1] - tr[:, i]
        print(f"  {nm:9s} mean {dt.mean():9.0f}  min {dt.min():9.0f}  max {dt.max():9.0f} ticks")
    whole = tr[:, 5] - tr[:, 0]
    print(f"  workgroup mean {whole.mean():9.0f}  max {whole.max():9.0f};  kernel span {float(tr[:, 5].max() - t0):9.0f} ticks")
    starts = (tr[:, 0] - t0).sort().values
    ends = (tr[:, 5] - t0).sort().values
    q = lambda v, f: float(v[int(f * (len(v) - 1))])
    print("  starts at ticks (quantiles 0 .5 .8 .9 1):", [round(q(starts, f)) for f in (0, .5, .8, .9, 1)])
    print("  ends   at ticks (quantiles 0 .5 .8 .9 1):", [round(q(ends, f)) for f in (0, .5, .8, .9, 1)])
    print("  loss", float(loss))


if __name__ == "__main__":
    main()
