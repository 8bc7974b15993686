#!/usr/bin/env python
"""Where does a wavefront of the WGAN critic's row kernel (k_critic_rows_mfma) spend its time?  Needs a tracing build:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGDA_CRITIC_TRACE -c pygda_amd/csrc/gda_critic.hip -o /tmp/critic_trace.o
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/critic_trace.o <the other objects of pygda_amd/csrc/build/ except gda_critic.o> -ldl -o <lib.so>
    python tools/critic_trace.py <lib.so> [n_s 9360] [n_t 8935] [h 128] [a 40] [dropout 0.4]

Calls gda_wgan_critic_f32 at bench.py's AdaGCN shapes (ACMv9 -> Citationv1 stand-ins; interpolates = twice the smaller
domain) a few times and prints, from
the clock stamps of every wavefront's first tile (lane 0), the mean / max cycles between the phase boundaries and when
the wavefronts started and ended."""
import ctypes
import sys

import torch

P = ctypes.c_void_p


def main():
    lib = ctypes.CDLL(sys.argv[1])
    n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 9360
    n_t = int(sys.argv[3]) if len(sys.argv) > 3 else 8935
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    a = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    p_drop = float(sys.argv[6]) if len(sys.argv) > 6 else 0.4
    n_i = n_t if n_s == n_t else 2 * min(n_s, n_t)          # adagcn.py:423-434: the smaller domain twice
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    es, et = torch.randn(n_s, h, generator=g).to(dev), (torch.randn(n_t, h, generator=g) + 0.3).to(dev)
    isrc = torch.randint(0, n_s, (n_i,), generator=g, dtype=torch.int32).to(dev)
    itgt = torch.randint(0, n_t, (n_i,), generator=g, dtype=torch.int32).to(dev)
    alpha = torch.rand(n_i, generator=g).to(dev)
    W1, b1 = (torch.randn(a, h, generator=g) * 0.1).to(dev), torch.zeros(a, device=dev)
    w2, b2 = (torch.randn(a, generator=g) * 0.1).to(dev), torch.zeros(1, device=dev)
    lib.gda_wgan_critic_workspace_bytes.restype = ctypes.c_size_t
    I64 = ctypes.c_int64
    wsb = lib.gda_wgan_critic_workspace_bytes(I64(n_s), I64(n_t), I64(n_i), h, a)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    loss, gW1, gb1, gw2, gb2 = (torch.empty(1, device=dev), torch.empty(a, h, device=dev), torch.empty(a, device=dev),
                                torch.empty(a, device=dev), torch.empty(1, device=dev))
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    waves = 512 * 4
    trace = torch.zeros(waves * 16, dtype=torch.int64, device=dev)
    assert lib.gda_dbg_critic_trace(P(trace.data_ptr())) == 0
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        trace.zero_()
        st = lib.gda_wgan_critic_f32(
            P(es.data_ptr()), I64(n_s), P(et.data_ptr()), I64(n_t), h, P(isrc.data_ptr()), P(itgt.data_ptr()),
            P(alpha.data_ptr()), I64(n_i), P(W1.data_ptr()), P(b1.data_ptr()), P(w2.data_ptr()), P(b2.data_ptr()), a,
            ctypes.c_float(p_drop), ctypes.c_uint64(1234), P(step.data_ptr()), ctypes.c_uint32(0), ctypes.c_float(10.0),
            P(loss.data_ptr()), P(gW1.data_ptr()), P(gb1.data_ptr()), P(gw2.data_ptr()), P(gb2.data_ptr()),
            P(ws.data_ptr()), ctypes.c_size_t(wsb), P(stream))
        assert st == 0, st
    torch.cuda.synchronize()
    tr = trace.view(-1, 16).cpu()
    # where the workgroups ran (slot 11: HW_ID | XCC_ID << 32): clocks agree within a CU only, so "two rounds" shows as
    # two workgroups on one CU whose [start, end] do not overlap
    hw = tr[:, 11]
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)
    wg = tr.view(-1, 4, 16)
    wg_ok = wg[:, :, 10].min(dim=1).values > 0
    cu_wg = cu.view(-1, 4)[:, 0]
    used = {}
    for b in range(len(wg)):
        if wg_ok[b]:
            used.setdefault(int(cu_wg[b]), []).append((int(wg[b, :, 0].min()), int(wg[b, :, 10].max())))
    multi = {k: sorted(v) for k, v in used.items() if len(v) > 1}
    wall = (wg[wg_ok][:, :, 13].max(dim=1).values - wg[wg_ok][:, :, 12].min(dim=1).values).double() * 0.01       # 100 MHz
    cyc = (wg[wg_ok][:, :, 10].max(dim=1).values - wg[wg_ok][:, :, 0].min(dim=1).values).double()
    print(f"  workgroup wall time mean {wall.mean():.1f} us max {wall.max():.1f} us; launch span (constant clock) "
          f"{(int(wg[wg_ok][:, :, 13].max()) - int(wg[wg_ok][:, :, 12].min())) * 0.01:.1f} us; shader clock {float((cyc / wall).mean()):.0f} MHz")
    print(f"  {int(wg_ok.sum())} workgroups on {len(used)} CUs; {len(multi)} CUs ran more than one")
    for k, v in list(multi.items())[:6]:
        print(f"    CU {k}: " + ", ".join(f"[{a - v[0][0]}, {b - v[0][0]}]" for a, b in v))
    tr = tr[tr[:, 10] > 0].double()
    print(f"{len(tr)} wavefronts stamped; loss {float(loss):.6f}")
    names = ["W1 -> LDS + sync", "X tile -> LDS", "Z = W1 X^T", "gap terms, (cAg U)^T X, reduce", "penalty masks",
             "U tile, V^T = W1^T U^T, Y", "T' = W1 Y^T", "U^T Y, column sums", "block reduce, partial out", "block partials"]
    for i, nm in enumerate(names):
        dt = tr[:, i + 1] - tr[:, i]
        print(f"  {nm:28s} mean {dt.mean():9.0f}  min {dt.min():9.0f}  max {dt.max():9.0f}")
    whole = tr[:, 10] - tr[:, 0]
    print(f"  wavefront mean {whole.mean():9.0f}  max {whole.max():9.0f}")


if __name__ == "__main__":
    main()
