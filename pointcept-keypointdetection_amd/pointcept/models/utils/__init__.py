from .misc import offset2batch, offset2bincount, bincount2offset, batch2offset, off_diagonal  # noqa: F401
from .serialization import encode  # noqa: F401
