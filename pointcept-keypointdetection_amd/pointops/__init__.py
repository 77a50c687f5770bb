"""`pointops` on MI355X: knn_query / grouping / interpolation of the reference's libs/pointops.

Same Python surface as libs/pointops/functions/__init__.py:1-14 for the three op families on the hot
path; `import pointops` is a hard import of the reference's trainer
hooks (engines/hooks/evaluator.py:12).  The other six families (ball query, FPS, subtraction,
aggregation, attention steps) serve PTv1/PTv2/Stratified-Transformer only and raise NotImplementedError.
"""
from .functions import (knn_query, grouping, grouping2, interpolation, interpolation2, knn_query_and_group,
                        offset2batch, batch2offset)  # noqa: F401
from . import _C  # noqa: F401


def _unsupported(name):
    def f(*a, **k):
        raise NotImplementedError(f"pointops.{name}: not part of the MI355X PTv3 path (SURVEY.md section 2b N1)")
    f.__name__ = name
    return f


for _n in ("ball_query", "random_ball_query", "farthest_point_sampling", "subtraction", "aggregation",
           "attention_relation_step", "attention_fusion_step", "query_and_group", "ball_query_and_group"):
    globals()[_n] = _unsupported(_n)
