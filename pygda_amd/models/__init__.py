from .base import BaseGDA
from .a2gnn import A2GNN
from .grade import GRADE
from .udagcn import UDAGCN
from .adagcn import AdaGCN
from .gnn import GNN
from .dane import DANE
from .tdss import TDSS
from .specreg import SpecReg
from .dgsda import DGSDA
from .strurw import StruRW

__all__ = ["BaseGDA", "A2GNN", "GRADE", "UDAGCN", "AdaGCN", "GNN", "DANE", "TDSS", "SpecReg", "DGSDA", "StruRW"]
