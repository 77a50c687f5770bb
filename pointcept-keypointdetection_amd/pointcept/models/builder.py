"""MODELS / MODULES registries and build_model (reference: pointcept/models/builder.py:12-17)."""
import copy

from pointcept.utils.registry import Registry

MODELS = Registry("models")
MODULES = Registry("modules")


def build_model(cfg):
    """Deep-copies cfg, pops "type", instantiates the registered class with the remaining keys."""
    return MODELS.build(copy.deepcopy(cfg))
