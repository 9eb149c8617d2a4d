from .network import MLP, NeRFNetwork, SkipConnMLP  # noqa: F401
from .renderer import NeRFRenderer, contract, near_far_from_aabb, sample_pdf  # noqa: F401
from .utils import collate_rays, get_rays  # noqa: F401
from .utils import freeze_loaded_parameters, load_checkpoint, save_checkpoint  # noqa: F401
from .sam_cache import SamFeatureCache, feature_map  # noqa: F401
