"""GPU: graph-classification mode (second half): ``A2GNN(mode='graph')`` -- DataLoader batches
of whole graphs, the mean readout of pygda/nn/a2gnn_base.py:140-141 as a segmented-mean kernel, the linear
classifier -- against goldens recorded from the reference's own a2gnn.py / a2gnn_base.py (through the PyG stub:
assumption 11 of tests/golden/_pyg_stub.py) and against the CPU oracle."""
import numpy as np
import pytest
import torch

import pygda_amd
from pygda_amd import ops
from pygda_amd.data import Data, DataLoader, collate_graphs
from oracle import pygda_cpu as O
from tests.conftest import T, load_golden, sub
from tests.test_gpu_parity import DEV, LOGIT_ATOL, REL, close, exact

pytestmark = pytest.mark.gpu


def _dataset(g, prefix):
    return [Data(x=T(g[f"{prefix}/{i}/x"]), edge_index=T(g[f"{prefix}/{i}/ei"]), y=T(g[f"{prefix}/{i}/y"]))
            for i in range(int(g[f"{prefix}/count"]))]


@pytest.mark.parametrize("d", [1, 16, 130])
def test_segment_mean_kernel_vs_oracle(d):
    """global_mean_pool forward bit-exact against the CPU scatter (rows added in node order, one division), the
    backward against autograd; graphs of one node, an EMPTY graph in the middle (count clamps at 1), ragged sizes;
    an unsorted batch vector is rejected."""
    gen = torch.Generator().manual_seed(d)
    counts = torch.tensor([3, 1, 0, 17, 40, 2, 0, 9])
    batch = torch.repeat_interleave(torch.arange(len(counts)), counts)
    x = torch.randn(int(counts.sum()), d, generator=gen)
    want = O.global_mean_pool(x, batch, len(counts))
    xd = x.to(DEV).requires_grad_()
    got = ops.segment_mean(xd, batch.to(DEV), len(counts))
    exact(got, want)
    assert bool((got[2] == 0).all()) and bool((got[6] == 0).all())
    gy = torch.randn(len(counts), d, generator=gen)
    got.backward(gy.to(DEV))
    xr = x.clone().requires_grad_()
    O.global_mean_pool(xr, batch, len(counts)).backward(gy)
    exact(xd.grad, xr.grad)
    # the graph count defaults to the last index + 1
    exact(ops.segment_mean(x.to(DEV), batch.to(DEV)), O.global_mean_pool(x, batch))
    with pytest.raises(ValueError):
        ops.segment_mean(x.to(DEV), batch.flip(0).contiguous().to(DEV), len(counts))


def test_collate_and_loader_follow_pyg():
    g = load_golden("a2gnn_graph_forward_mmd")
    ds = _dataset(g, "src")
    b = collate_graphs(ds)
    ob = O.collate_graphs([O.Graph(d.x, d.edge_index, d.y) for d in ds])
    exact(b.x, ob.x); exact(b.edge_index, ob.edge_index); exact(b.y, ob.y); exact(b.batch, ob.batch)
    assert b.num_graphs == len(ds) == 14
    # torch's own loader: the same shuffles as a torch DataLoader built anywhere else from the same generator state
    import torch.utils.data as tud
    torch.manual_seed(3)
    mine = [bb.y.tolist() for bb in DataLoader(ds, batch_size=5, shuffle=True)]
    torch.manual_seed(3)
    ref = [bb.y.tolist() for bb in tud.DataLoader(ds, batch_size=5, shuffle=True, collate_fn=collate_graphs)]
    assert mine == ref and [len(m) for m in mine] == [5, 5, 4]


def test_a2gnn_graph_mode_forward_model_golden():
    g = load_golden("a2gnn_graph_forward_mmd")
    sb, tb = collate_graphs(_dataset(g, "src")), collate_graphs(_dataset(g, "tgt"))
    m = pygda_amd.models.A2GNN(10, 16, 3, mode='graph', num_layers=2, dropout=0.0, s_pnums=0, t_pnums=5, weight=0.5,
                               device=DEV, epoch=3, verbose=0)
    torch.manual_seed(int(g["init_seed"]))
    m.a2gnn = m.init_model()
    for k, v in sub(g, "param/").items():
        exact(m.a2gnn.state_dict()[k], v)
    m.a2gnn.train()
    sd, td = sb.to(DEV), tb.to(DEV)
    close(m.a2gnn.feat_bottleneck(sd.x, sd.edge_index, sd.batch, 0), g["pooled_src"], rtol=0, atol=1e-5)
    torch.manual_seed(int(g["mmd_seed"]))
    loss, sl, tl = m.forward_model(sd, td, float(g["alpha"]))
    loss.backward()
    close(loss, g["loss"], rtol=REL)
    close(sl, g["src_logits"], rtol=0, atol=LOGIT_ATOL); close(tl, g["tgt_logits"], rtol=0, atol=LOGIT_ATOL)
    params = dict(m.a2gnn.named_parameters())
    for k, v in sub(g, "grad/").items():
        close(params[k].grad, v, rtol=1e-3, atol=1e-4 * max(np.abs(v).max(), 1e-3))
    # the reference cannot run its adversarial branch in graph mode (labels sized by nodes): the same ValueError
    ma = pygda_amd.models.A2GNN(10, 16, 3, mode='graph', num_layers=2, dropout=0.0, s_pnums=0, t_pnums=5, adv=True,
                                device=DEV, epoch=1, verbose=0)
    ma.a2gnn = ma.init_model()
    with pytest.raises(ValueError):
        ma.forward_model(sd, td, 0.5)


@pytest.mark.parametrize("batch_size", [0, 6])
def test_a2gnn_graph_mode_fit_predict_golden(batch_size):
    """fit() for three epochs from the reference's seed over shuffled DataLoader batches, then predict()."""
    g = load_golden(f"a2gnn_graph_fit3_b{batch_size}")
    src, tgt = _dataset(g, "src"), _dataset(g, "tgt")
    m = pygda_amd.models.A2GNN(10, 16, 3, mode='graph', num_layers=2, dropout=0.0, s_pnums=0, t_pnums=5, weight=0.5,
                               lr=0.01, weight_decay=0.001, device=DEV, epoch=3, verbose=0, batch_size=batch_size)
    seen = []
    m.epoch_hook = lambda e, loss, acc, secs: seen.append((loss, acc))
    torch.manual_seed(int(g["seed"]))
    m.fit(src, tgt)
    close([x[0] for x in seen], g["losses"], rtol=REL)
    close([x[1] for x in seen], g["accs"], rtol=0, atol=1e-12)
    for k, v in sub(g, "final/").items():
        close(m.a2gnn.state_dict()[k], v, rtol=1e-3, atol=2e-4)
    if batch_size == 0:
        logits, labels = m.predict(tgt)          # draws the loader's next shuffle, like the reference's predict()
        close(logits, g["tgt_logits"], rtol=0, atol=LOGIT_ATOL)
        exact(labels, g["tgt_labels"])
        exact(logits.argmax(1), g["tgt_logits"].argmax(1))
