from .point_transformer_v3m1_base import *  # noqa: F401,F403
from .point_transformer_v3m2_sonata import PointTransformerV3 as PointTransformerV3m2  # noqa: F401,E402
