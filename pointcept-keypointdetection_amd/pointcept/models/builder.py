"""Model registries of the package (interface of the reference's pointcept/models/builder.py:12-17):
`MODELS` holds backbones and wrappers, `MODULES` reusable parts; `build_model(cfg)` instantiates
`cfg["type"]` from `MODELS` with the remaining keys and never mutates the caller's config."""
from copy import deepcopy

from pointcept.utils.registry import Registry

MODELS, MODULES = Registry("models"), Registry("modules")


def build_model(cfg):
    return MODELS.build(deepcopy(cfg))
