from .mmd import MMD, get_MMD, guassian_kernel
from .utility import logger

__all__ = ["MMD", "get_MMD", "guassian_kernel", "logger"]
