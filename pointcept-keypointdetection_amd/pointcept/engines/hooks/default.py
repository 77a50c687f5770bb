"""HookBase (reference: pointcept/engines/hooks/default.py:13-36)."""
from . import HookBase  # noqa: F401
