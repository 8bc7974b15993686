"""CPU: the trainer / loader logic of ``mode='graph'`` (GRADE, UDAGCN, AdaGCN, DANE -- SURVEY 8 f4) against the goldens
recorded from the reference's own files (tests/golden/graph_trainers.npz), with the CPU oracle injected UNDER the operator
layer: the product's operators have no CPU path (they raise), so the numeric leaves -- aggregation, graph ingestion,
readout, fused losses -- are the oracle's here, and what this file checks is everything above them: loaders and their
shuffles, the order of every CPU-generator draw, pooling call sites, label / row counts, optimisers, the epoch loop,
predict().  The same goldens are checked on the HIP kernels in tests/test_gpu_graph_mode.py."""
import contextlib

import numpy as np
import pytest
import torch

import pygda_amd
from oracle import pygda_cpu as O
from pygda_amd.data import Data
from tests.conftest import T, load_golden, sub


@contextlib.contextmanager
def oracle_leaves():
    from tests.dp_equality import inject_oracle
    import pygda_amd.models.adagcn as MA
    import pygda_amd.models.dane as MD
    import pygda_amd.models.grade as MG
    import pygda_amd.models.udagcn as MU
    import pygda_amd.nn.a2gnn_base as AB
    import pygda_amd.nn.adagcn_base as ADB
    import pygda_amd.nn.gnn_base as GB
    import pygda_amd.nn.grade_base as GRB
    restore = inject_oracle()

    def pool(x, batch, size=None):
        return O.global_mean_pool(x, batch, size)

    def grl_disc_ce(fs, ft, W, b, alpha, labels=None):              # grade.py:169-176 on the oracle's GradReverse
        z = torch.nn.functional.linear(O.grad_reverse(torch.cat([fs, ft]), float(alpha)), W, b)
        y = torch.cat([torch.zeros(fs.size(0), dtype=torch.long), torch.ones(ft.size(0), dtype=torch.long)])
        return torch.nn.functional.cross_entropy(z, y)

    saved = [(mod, "global_mean_pool", mod.global_mean_pool) for mod in (MU, MD, AB, ADB, GB, GRB)]
    saved += [(MG, "grl_disc_ce", MG.grl_disc_ce), (MG, "MMD", MG.MMD)]
    for mod, name, _ in saved[:-2]:
        setattr(mod, name, pool)
    MG.grl_disc_ce = grl_disc_ce
    MG.MMD = lambda s, t: O.MMD(s, t)
    try:
        yield
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
        restore()


def _dataset(g, prefix):
    return [Data(x=T(g[f"{prefix}/{i}/x"]), edge_index=T(g[f"{prefix}/{i}/ei"]), y=T(g[f"{prefix}/{i}/y"]))
            for i in range(int(g[f"{prefix}/count"]))]


def _trainer(tag, **kw):
    M = pygda_amd.models
    common = dict(device="cpu", epoch=3, verbose=0, **kw)
    if tag.startswith("grade"):
        return M.GRADE(10, 8, 3, mode='graph', num_layers=2, dropout=0.0, disc=tag[6:].upper(), weight=0.5, lr=0.01,
                       weight_decay=0.001, **common), "grade"
    if tag == "udagcn":
        return M.UDAGCN(10, 8, 3, mode='graph', num_layers=2, ppmi=False, adv_dim=6, lr=0.01, weight_decay=0.003,
                        **common), "udagcn"
    if tag == "adagcn":
        return M.AdaGCN(10, 8, 3, mode='graph', num_layers=2, adv_dim=6, gp_weight=5, domain_weight=1, lr=0.01,
                        weight_decay=0.001, **common), "adagcn"
    return M.DANE(10, 8, 3, num_layers=2, mode='graph', dropout=0.0, gnn="gcn", k=5, lr=0.01, weight_decay=1e-5,
                  **common), "gnn"


@pytest.mark.parametrize("batch_size", [0, 6])
@pytest.mark.parametrize("tag", ["grade_js", "grade_mmd", "udagcn", "adagcn", "dane"])
def test_graph_mode_fit_predict_host_logic(monkeypatch, tag, batch_size):
    import torch.nn as nn
    orig = nn.Dropout.__init__
    monkeypatch.setattr(nn.Dropout, "__init__", lambda self, p=0.5, inplace=False: orig(self, 0.0, inplace))
    g = load_golden("graph_trainers")
    src, tgt = _dataset(g, "src"), _dataset(g, "tgt")
    with oracle_leaves():
        m, attr = _trainer(tag, batch_size=batch_size)
        seen = []
        m.epoch_hook = lambda e, loss, acc, secs: seen.append((float(loss), acc))
        torch.manual_seed(int(g["fit_seed"]))
        m.fit(src, tgt)
        np.testing.assert_allclose([x[0] for x in seen], g[f"{tag}/fit{batch_size}/losses"], rtol=1e-5)
        np.testing.assert_allclose([x[1] for x in seen], g[f"{tag}/fit{batch_size}/accs"], atol=1e-12)
        net = getattr(m, attr)
        for k, v in sub(g, f"{tag}/fit{batch_size}/final/").items():
            np.testing.assert_allclose(net.state_dict()[k].numpy(), v, rtol=1e-4, atol=1e-5)
        if batch_size == 0:
            logits, labels = m.predict(tgt)
            np.testing.assert_allclose(logits.numpy(), g[f"{tag}/fit0/tgt_logits"], atol=1e-5)
            np.testing.assert_array_equal(labels.numpy(), g[f"{tag}/fit0/tgt_labels"])
