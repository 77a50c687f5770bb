"""String -> class registry with the mmcv-style surface the reference's callers use.

Mirrors the behaviour of pointcept/utils/registry.py (reference): `Registry.register_module(name=None,
force=False, module=None)` as decorator or call, `get`, `build(cfg)`; `build_from_cfg` pops "type",
instantiates with the remaining keys and re-raises constructor errors as
`type(e)(f"{cls.__name__}: {e}")` (registry.py:9-56); re-registering a name raises KeyError unless
force=True (registry.py:238-249).
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and (default_args is None or "type" not in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be a Registry object, but got {type(registry)}")
    if not (isinstance(default_args, dict) or default_args is None):
        raise TypeError(f"default_args must be a dict or None, but got {type(default_args)}")
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return obj_cls(**args)
    except Exception as e:  # plain TypeErrors do not name the class
        raise type(e)(f"{obj_cls.__name__}: {e}")


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self._children = {}
        self._scope = scope if scope is not None else "pointcept"
        self.parent = parent
        if build_func is not None:
            self.build_func = build_func
        elif parent is not None:
            self.build_func = parent.build_func
        else:
            self.build_func = build_from_cfg
        if parent is not None:
            assert isinstance(parent, Registry)
            parent._children[self._scope] = self

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self._name}, items={self._module_dict})"

    @property
    def name(self):
        return self._name

    @property
    def scope(self):
        return self._scope

    @property
    def module_dict(self):
        return self._module_dict

    @property
    def children(self):
        return self._children

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f"module must be a class, but got {type(module_class)}")
        names = [module_class.__name__] if module_name is None else (
            [module_name] if isinstance(module_name, str) else list(module_name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self.name}")
            self._module_dict[n] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f"force must be a boolean, but got {type(force)}")
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f"name must be None, a str or a sequence of str, but got {type(name)}")
        if module is not None:
            self._register_module(module_class=module, module_name=name, force=force)
            return module

        def _register(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls

        return _register
