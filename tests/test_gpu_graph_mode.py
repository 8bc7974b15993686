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


# ---- mode='graph' of GRADE / UDAGCN / AdaGCN / DANE (tests/golden/graph_trainers.npz) ------------------
def _gt_trainer(tag, **kw):
    M = pygda_amd.models
    common = dict(device=DEV, epoch=3, verbose=0, **kw)
    if tag == "grade_js":
        return M.GRADE(10, 8, 3, mode='graph', num_layers=2, dropout=0.0, disc="JS", weight=0.5, lr=0.01,
                       weight_decay=0.001, **common), "grade"
    if tag == "grade_mmd":
        return M.GRADE(10, 8, 3, mode='graph', num_layers=2, dropout=0.0, disc="MMD", weight=0.5, lr=0.01,
                       weight_decay=0.001, **common), "grade"
    if tag == "udagcn":
        return M.UDAGCN(10, 8, 3, mode='graph', num_layers=2, ppmi=False, adv_dim=6, lr=0.01, weight_decay=0.003,
                        **common), "udagcn"
    if tag == "adagcn":
        return M.AdaGCN(10, 8, 3, mode='graph', num_layers=2, adv_dim=6, gp_weight=5, domain_weight=1, lr=0.01,
                        weight_decay=0.001, **common), "adagcn"
    return M.DANE(10, 8, 3, num_layers=2, mode='graph', dropout=0.0, gnn="gcn", k=5, lr=0.01, weight_decay=1e-5,
                  **common), "gnn"


def _zero_dropouts(monkeypatch):
    import torch.nn as nn
    orig = nn.Dropout.__init__
    monkeypatch.setattr(nn.Dropout, "__init__", lambda self, p=0.5, inplace=False: orig(self, 0.0, inplace))


@pytest.mark.parametrize("tag", ["grade_js", "grade_mmd", "udagcn", "adagcn", "dane"])
def test_graph_mode_forward_model_golden(monkeypatch, tag):
    """One ``forward_model`` on the collated datasets against the reference's own files run in graph mode (grade.py,
    grade_base.py:150-157; udagcn.py:168-170; adagcn.py + adagcn_base.py:93-94; dane.py:171-176, 323-331, 448-456,
    492-493): loss, logits, every gradient; for AdaGCN / DANE the critic / discriminator (and DANE's encoder) after
    their own Adam steps."""
    import torch.nn as nn
    _zero_dropouts(monkeypatch)
    g = load_golden("graph_trainers")
    sb, tb = collate_graphs(_dataset(g, "src")).to(DEV), collate_graphs(_dataset(g, "tgt")).to(DEV)
    m, attr = _gt_trainer(tag)
    torch.manual_seed(int(g["init_seed"]))
    net = m.init_model()
    setattr(m, attr, net)
    for k, v in sub(g, f"{tag}/param/").items():
        exact(net.state_dict()[k], v)
    net.train()
    for mod in getattr(net, "models", []):
        mod.train()
    if tag == "adagcn":
        m.discriminator = nn.Sequential(nn.Linear(8, 6), nn.ReLU(), nn.Dropout(0.1), nn.Linear(6, 1), nn.Sigmoid()).to(DEV)
        m.discriminator.load_state_dict({k: T(v) for k, v in sub(g, f"{tag}/disc0/").items()})
        m.c_optimizer = torch.optim.Adam(m.discriminator.parameters(), lr=0.01, weight_decay=0.001)
    if tag == "dane":
        m.domain_discriminator = nn.Sequential(nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 1)).to(DEV)
        m.domain_discriminator.load_state_dict({k: T(v) for k, v in sub(g, f"{tag}/disc0/").items()})
        m.sample_size = min(int(g["src/count"]), int(g["tgt/count"]))
        m.g_optimizer = torch.optim.Adam(net.parameters(), lr=0.01, weight_decay=1e-5)
        m.d_optimizer = torch.optim.Adam(m.domain_discriminator.parameters(), lr=0.01, weight_decay=1e-5)
    torch.manual_seed(int(g["draw_seed"]))
    if tag.startswith("grade"):
        loss, sl, tl = m.forward_model(sb, tb, 0.4)
    elif tag == "udagcn":
        loss, sl, tl = m.forward_model(sb, tb, 0.05, 2)
    else:
        loss, sl, tl = m.forward_model(sb, tb)
    assert sl.shape == (int(g["src/count"]), 3) and tl.shape == (int(g["tgt/count"]), 3)      # one row per graph
    close(sl, g[f"{tag}/src_logits"], rtol=0, atol=LOGIT_ATOL); close(tl, g[f"{tag}/tgt_logits"], rtol=0, atol=LOGIT_ATOL)
    close(float(loss), float(g[f"{tag}/loss"]), rtol=REL)
    if tag == "dane":
        for k, v in sub(g, f"{tag}/param1/").items():
            close(net.state_dict()[k], v, rtol=1e-3, atol=2e-4)
        for k, v in sub(g, f"{tag}/disc1/").items():
            close(m.domain_discriminator.state_dict()[k], v, rtol=1e-3, atol=2e-4)
        return
    net.zero_grad()
    loss.backward()
    params = dict(net.named_parameters())
    want = sub(g, f"{tag}/grad/")
    assert want
    for k, v in want.items():
        close(params[k].grad, v, rtol=1e-3, atol=1e-4 * max(np.abs(v).max(), 1e-3))
    if tag == "adagcn":
        for k, v in sub(g, f"{tag}/disc10/").items():
            close(m.discriminator.state_dict()[k], v, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("batch_size", [0, 6])
@pytest.mark.parametrize("tag", ["grade_js", "grade_mmd", "udagcn", "adagcn", "dane"])
def test_graph_mode_fit_predict_golden(monkeypatch, tag, batch_size):
    """fit() for three epochs from the reference's seed over shuffled DataLoader batches (one batch per domain, and
    mini-batches of six graphs), then predict() for the one-batch run."""
    _zero_dropouts(monkeypatch)
    g = load_golden("graph_trainers")
    src, tgt = _dataset(g, "src"), _dataset(g, "tgt")
    m, attr = _gt_trainer(tag, batch_size=batch_size)
    seen = []
    m.epoch_hook = lambda e, loss, acc, secs: seen.append((float(loss), acc))
    torch.manual_seed(int(g["fit_seed"]))
    m.fit(src, tgt)
    close([x[0] for x in seen], g[f"{tag}/fit{batch_size}/losses"], rtol=REL)
    close([x[1] for x in seen], g[f"{tag}/fit{batch_size}/accs"], rtol=0, atol=1e-12)
    net = getattr(m, attr)
    for k, v in sub(g, f"{tag}/fit{batch_size}/final/").items():
        close(net.state_dict()[k], v, rtol=1e-3, atol=2e-4)
    if batch_size == 0:
        logits, labels = m.predict(tgt)          # draws the loader's next shuffle, like the reference's predict()
        close(logits, g[f"{tag}/fit0/tgt_logits"], rtol=0, atol=LOGIT_ATOL)
        exact(labels, g[f"{tag}/fit0/tgt_labels"])
        exact(logits.argmax(1), g[f"{tag}/fit0/tgt_logits"].argmax(1))
