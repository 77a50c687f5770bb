from .builder import LOSSES, Criteria, build_criteria  # noqa: F401
from .misc import CrossEntropyLoss  # noqa: F401
