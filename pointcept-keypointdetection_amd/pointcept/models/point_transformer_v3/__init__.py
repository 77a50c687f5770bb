from .point_transformer_v3m1_base import *  # noqa: F401,F403
