"""Input side of the path on the device: GridSample / Collect and the collate function."""
from .transform import TRANSFORMS, GridSample, Collect, ToTensor, index_operator  # noqa: F401
from .utils import collate_fn, point_collate_fn  # noqa: F401
