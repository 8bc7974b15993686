from .mmd import MMD, get_MMD, guassian_kernel
from .utility import logger
from .svd_transform import svd_transform

__all__ = ["MMD", "get_MMD", "guassian_kernel", "logger", "svd_transform"]
