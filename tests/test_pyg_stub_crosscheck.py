"""Optional cross-check of the test-only PyG stub (tests/golden/_pyg_stub.py: the fourteen numbered assumptions every
reference-run golden rests on) against a REAL torch_geometric, where one is importable.  Neither this
container nor the GPU box has PyG -- no network to install it -- so everywhere the project's own suite runs these tests
skip; on a machine with PyG >= 2.4 they pin the stub's reading of PyG's initialisation order and operator semantics to
PyG itself: same seeded parameters, same outputs on a small directed graph with duplicate edges, self loops and an
isolated node."""
import importlib.util
import os

import pytest
import torch

pyg = pytest.importorskip("torch_geometric", reason="torch_geometric is not installed: the stub is the only PyG here")


def _stub():
    path = os.path.join(os.path.dirname(__file__), "golden", "_pyg_stub.py")
    spec = importlib.util.spec_from_file_location("_pyg_stub_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _graph():
    gen = torch.Generator().manual_seed(5)
    n, e = 40, 160
    ei = torch.randint(0, n - 1, (2, e), generator=gen)          # node n - 1 stays isolated
    ei = torch.cat([ei, ei[:, :7], torch.tensor([[3, 9], [3, 9]])], dim=1)      # duplicates and two self loops
    return n, ei, torch.randn(n, 12, generator=gen)


@pytest.mark.parametrize("name", ["GCNConv", "SAGEConv", "GATConv"])
def test_stub_convs_equal_pyg(name):
    from torch_geometric import nn as pnn
    stub = _stub()
    n, ei, x = _graph()
    torch.manual_seed(11)
    theirs = getattr(pnn, name)(12, 8)
    torch.manual_seed(11)
    ours = getattr(stub, name)(12, 8)
    sd_t, sd_o = theirs.state_dict(), ours.state_dict()
    assert sorted(sd_t) == sorted(sd_o), (sorted(sd_t), sorted(sd_o))
    for k in sd_t:
        assert torch.equal(sd_t[k], sd_o[k]), k                   # the same draws in the same order
    assert torch.allclose(theirs(x, ei), ours(x, ei), atol=1e-6)
    torch.manual_seed(11)
    getattr(pnn, name)(12, 8)
    a = torch.rand(4)
    torch.manual_seed(11)
    getattr(stub, name)(12, 8)
    assert torch.equal(a, torch.rand(4))                          # both constructors leave the generator at the same place


def test_stub_gin_and_utilities_equal_pyg():
    from torch_geometric import nn as pnn
    from torch_geometric import utils as putils
    stub = _stub()
    n, ei, x = _graph()

    def mlp():
        return torch.nn.Sequential(torch.nn.Linear(12, 8))
    torch.manual_seed(3)
    theirs = pnn.GINConv(mlp())
    torch.manual_seed(3)
    ours = stub.GINConv(mlp())
    for (ka, va), (kb, vb) in zip(sorted(theirs.state_dict().items()), sorted(ours.state_dict().items())):
        assert ka == kb and torch.equal(va, vb), ka
    assert torch.allclose(theirs(x, ei), ours(x, ei), atol=1e-6)
    w = torch.rand(ei.size(1))
    a_ei, a_w = putils.add_remaining_self_loops(ei, w, 2.0, n)
    b_ei, b_w = stub.add_remaining_self_loops(ei, w, 2.0, n)
    assert torch.equal(a_ei, b_ei) and torch.equal(a_w, b_w)
    assert torch.equal(putils.to_undirected(ei, num_nodes=n), stub.to_undirected(ei, num_nodes=n))
    batch = torch.sort(torch.randint(0, 5, (n,))).values
    assert torch.allclose(pnn.global_mean_pool(x, batch), stub.global_mean_pool(x, batch), atol=1e-6)
