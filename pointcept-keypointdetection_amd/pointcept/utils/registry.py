"""Name -> class registry with the surface the reference's callers use (pointcept/utils/registry.py, mmcv style).

Contract kept (and tested in tests/test_boundary_cpu.py): `Registry(name, build_func=None, parent=None, scope=None)`;
`register_module(name=None, force=False, module=None)` as a decorator or a direct call, with a string or a sequence of
strings as name(s); `get(key)` falls back to the parent; `key in registry`; `len(registry)`; `build(cfg, ...)` hands
over to `build_func` (default `build_from_cfg`), which pops "type" from a COPY of cfg, looks the class up (or takes a
class object as is), fills in `default_args`, instantiates, and re-raises constructor failures prefixed with the class
name as `type(e)(f"{cls.__name__}: {e}")` (registry.py:9-56).  A name registered twice raises KeyError unless
force=True (registry.py:238-249).  When the overlay is grafted into a reference checkout the reference's own file is
the one on the import path; this one serves the standalone package.
"""
import inspect


def _expect(ok, exc, message):
    if not ok:
        raise exc(message)


def build_from_cfg(cfg, registry, default_args=None):
    _expect(isinstance(cfg, dict), TypeError, f"cfg must be a dict, but got {type(cfg)}")
    defaults = default_args if default_args is not None else {}
    _expect(isinstance(defaults, dict), TypeError, f"default_args must be a dict or None, but got {type(default_args)}")
    _expect("type" in cfg or "type" in defaults, KeyError,
            f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    _expect(isinstance(registry, Registry), TypeError, f"registry must be a Registry object, but got {type(registry)}")
    kwargs = {**defaults, **cfg}          # cfg wins over the defaults; the caller's dicts stay untouched
    target = kwargs.pop("type")
    if inspect.isclass(target):
        cls = target
    else:
        _expect(isinstance(target, str), TypeError, f"type must be a str or valid type, but got {type(target)}")
        cls = registry.get(target)
        _expect(cls is not None, KeyError, f"{target} is not in the {registry.name} registry")
    try:
        return cls(**kwargs)
    except Exception as err:   # a bare TypeError from __init__ does not say which class it was
        raise type(err)(f"{cls.__name__}: {err}")


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        if parent is not None and not isinstance(parent, Registry):
            raise AssertionError("parent must be a Registry")
        self._name, self._scope = name, ("pointcept" if scope is None else scope)
        self._module_dict, self._children = {}, {}
        self.parent = parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)
        if parent is not None:
            parent._children[self._scope] = self

    name = property(lambda self: self._name)
    scope = property(lambda self: self._scope)
    module_dict = property(lambda self: self._module_dict)
    children = property(lambda self: self._children)

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={self._module_dict})"

    def get(self, key):
        node = self
        while node is not None:
            if key in node._module_dict:
                return node._module_dict[key]
            node = node.parent
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register_module(self, module_class, module_name=None, force=False):
        _expect(inspect.isclass(module_class), TypeError, f"module must be a class, but got {type(module_class)}")
        if module_name is None:
            module_name = module_class.__name__
        for key in ([module_name] if isinstance(module_name, str) else list(module_name)):
            _expect(force or key not in self._module_dict, KeyError, f"{key} is already registered in {self.name}")
            self._module_dict[key] = module_class

    def register_module(self, name=None, force=False, module=None):
        _expect(isinstance(force, bool), TypeError, f"force must be a boolean, but got {type(force)}")
        named = name is None or isinstance(name, str) or (
            isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))
        _expect(named, TypeError, f"name must be None, a str or a sequence of str, but got {type(name)}")

        def add(cls):
            self._register_module(module_class=cls, module_name=name, force=force)
            return cls

        return add if module is None else add(module)
