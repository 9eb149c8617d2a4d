from .network import MLP, NeRFNetwork, SkipConnMLP  # noqa: F401
from .renderer import NeRFRenderer, contract, near_far_from_aabb, sample_pdf  # noqa: F401
from .utils import get_rays  # noqa: F401
