from .base import BaseGDA
from .a2gnn import A2GNN
from .grade import GRADE

__all__ = ["BaseGDA", "A2GNN", "GRADE"]
