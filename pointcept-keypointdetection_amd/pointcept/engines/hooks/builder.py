"""Hook registry (reference: pointcept/engines/hooks/builder.py:8-17)."""
from pointcept.utils.registry import Registry

HOOKS = Registry("hooks")


def build_hooks(cfg):
    return [HOOKS.build(hook_cfg) for hook_cfg in cfg]
