"""Load individual files of the reference checkout by path (build container only).

``import pygda`` cannot work (its ``__init__`` pulls in torch_geometric datasets), so
a bare package skeleton ``pygda`` / ``pygda.nn`` / ``pygda.utils`` / ``pygda.models`` /
``pygda.metrics`` is registered and only the hot-path files are executed into it, in
dependency order, against ``_pyg_stub``.  Nothing is copied: the files run from
where they lie, with bytecode writing disabled so the read-only tree stays clean.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True

REF = os.environ.get("PYGDA_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "pygda", "utils", "mmd.py"))


def _pkg(name):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, *name.split("."))]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    if modname in sys.modules:
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    parent, _, leaf = modname.rpartition(".")
    setattr(sys.modules[parent], leaf, m)
    return m


def load_reference(with_pyg_stub=True):
    """Returns a namespace of reference symbols on the hot path."""
    if with_pyg_stub:
        from . import _pyg_stub
        _pyg_stub.install()
    for p in ("pygda", "pygda.nn", "pygda.utils", "pygda.models", "pygda.metrics"):
        _pkg(p)
    ns = types.SimpleNamespace()
    # pure-torch files (true oracle, no stub involved)
    mmd = _load("pygda.utils.mmd", "pygda/utils/mmd.py")
    util = _load("pygda.utils.utility", "pygda/utils/utility.py")
    rev = _load("pygda.nn.reverse_layer", "pygda/nn/reverse_layer.py")
    att = _load("pygda.nn.attention", "pygda/nn/attention.py")
    met = _load("pygda.metrics.metrics", "pygda/metrics/metrics.py")
    U, NN, M, MET = (sys.modules[k] for k in ("pygda.utils", "pygda.nn", "pygda.models", "pygda.metrics"))
    U.MMD, U.get_MMD, U.guassian_kernel, U.logger = mmd.MMD, mmd.get_MMD, mmd.guassian_kernel, util.logger
    NN.GradReverse, NN.Attention = rev.GradReverse, att.Attention
    MET.eval_micro_f1, MET.eval_macro_f1 = met.eval_micro_f1, met.eval_macro_f1
    ns.MMD, ns.get_MMD, ns.guassian_kernel = mmd.MMD, mmd.get_MMD, mmd.guassian_kernel
    ns.GradReverse, ns.Attention, ns.logger = rev.GradReverse, att.Attention, util.logger
    ns.eval_micro_f1, ns.eval_macro_f1 = met.eval_micro_f1, met.eval_macro_f1
    if not with_pyg_stub:
        return ns
    prop = _load("pygda.nn.prop_gcn_conv", "pygda/nn/prop_gcn_conv.py")
    cached = _load("pygda.nn.cached_gcn_conv", "pygda/nn/cached_gcn_conv.py")
    ppmi = _load("pygda.nn.ppmi_conv", "pygda/nn/ppmi_conv.py")
    a2b = _load("pygda.nn.a2gnn_base", "pygda/nn/a2gnn_base.py")
    grb = _load("pygda.nn.grade_base", "pygda/nn/grade_base.py")
    udb = _load("pygda.nn.udagcn_base", "pygda/nn/udagcn_base.py")
    adb = _load("pygda.nn.adagcn_base", "pygda/nn/adagcn_base.py")
    gnb = _load("pygda.nn.gnn_base", "pygda/nn/gnn_base.py")
    NN.GNNBase = gnb.GNNBase
    ns.GNNBase = gnb.GNNBase
    NN.PropGCNConv, NN.CachedGCNConv, NN.PPMIConv = prop.PropGCNConv, cached.CachedGCNConv, ppmi.PPMIConv
    NN.A2GNNBase, NN.GRADEBase, NN.UDAGCNBase, NN.AdaGCNBase = a2b.A2GNNBase, grb.GRADEBase, udb.UDAGCNBase, adb.AdaGCNBase
    base = _load("pygda.models.base", "pygda/models/base.py")
    M.BaseGDA = base.BaseGDA
    a2 = _load("pygda.models.a2gnn", "pygda/models/a2gnn.py")
    gr = _load("pygda.models.grade", "pygda/models/grade.py")
    ud = _load("pygda.models.udagcn", "pygda/models/udagcn.py")
    ad = _load("pygda.models.adagcn", "pygda/models/adagcn.py")
    gn = _load("pygda.models.gnn", "pygda/models/gnn.py")
    da = _load("pygda.models.dane", "pygda/models/dane.py")
    ns.GNN, ns.DANE = gn.GNN, da.DANE
    td = _load("pygda.models.tdss", "pygda/models/tdss.py")
    ns.TDSS, ns.TwoHopNeighbor = td.TDSS, td.TwoHopNeighbor
    dgb = _load("pygda.nn.dgsda_base", "pygda/nn/dgsda_base.py")
    NN.DGSDABase = dgb.DGSDABase
    dg = _load("pygda.models.dgsda", "pygda/models/dgsda.py")
    ns.BernProp, ns.DGSDABase, ns.DGSDA = dgb.BernProp, dgb.DGSDABase, dg.DGSDA
    rwg = _load("pygda.nn.reweight_gnn", "pygda/nn/reweight_gnn.py")
    mxc = _load("pygda.nn.mixup_gcnconv", "pygda/nn/mixup_gcnconv.py")
    NN.MixUpGCNConv = mxc.MixUpGCNConv
    mxb = _load("pygda.nn.mixup_base", "pygda/nn/mixup_base.py")
    NN.ReweightGNN, NN.MixupBase = rwg.ReweightGNN, mxb.MixupBase
    ns.MixUpGCNConv, ns.MixupBase = mxc.MixUpGCNConv, mxb.MixupBase
    stw = _load("pygda.models.strurw", "pygda/models/strurw.py")
    ns.ReweightGNN, ns.GCN_reweight, ns.GS_reweight, ns.StruRW = rwg.ReweightGNN, rwg.GCN_reweight, rwg.GS_reweight, stw.StruRW
    sr = _load("pygda.models.specreg", "pygda/models/specreg.py")
    ns.SpecReg = sr.SpecReg
    ns.gcn_norm, ns.PropGCNConv = prop.gcn_norm, prop.PropGCNConv
    ns.CachedGCNConv, ns.PPMIConv = cached.CachedGCNConv, ppmi.PPMIConv
    ns.A2GNNBase, ns.GRADEBase, ns.UDAGCNBase, ns.AdaGCNBase = a2b.A2GNNBase, grb.GRADEBase, udb.UDAGCNBase, adb.AdaGCNBase
    ns.BaseGDA, ns.A2GNN, ns.GRADE, ns.UDAGCN, ns.AdaGCN = base.BaseGDA, a2.A2GNN, gr.GRADE, ud.UDAGCN, ad.AdaGCN
    return ns
