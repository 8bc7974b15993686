"""Drop-in alias: ``import pygda`` / ``from pygda.models import A2GNN`` resolve to the
MI355X-native implementation in :mod:`pygda_amd`, so scripts written against pygda-team/pygda
(benchmark/node/*.py, examples/demo.py) run unchanged for the trainers this build covers:
A2GNN, GRADE, UDAGCN, AdaGCN, DANE, GNN, TDSS, SpecReg, DGSDA, StruRW and the ``pygda.nn`` operators they use."""
import sys

import pygda_amd
from pygda_amd import datasets, metrics, models, nn, utils

for _name in ("nn", "models", "utils", "metrics", "datasets"):
    sys.modules[__name__ + "." + _name] = getattr(pygda_amd, _name)
for _pkg in ("nn", "models", "utils", "metrics", "datasets"):     # submodules: pygda.nn.prop_gcn_conv, ...
    _prefix = "pygda_amd." + _pkg + "."
    for _k, _v in list(sys.modules.items()):
        if _k.startswith(_prefix):
            sys.modules[__name__ + "." + _pkg + "." + _k[len(_prefix):]] = _v

__version__ = pygda_amd.__version__
