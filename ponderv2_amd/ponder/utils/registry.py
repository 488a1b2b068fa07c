"""Name -> class registries with ``build(cfg)`` from a ``dict(type=..., **kwargs)``.

API-compatible with the subset of ponder/utils/registry.py the hot path uses
(``Registry.register_module`` :262-316, ``Registry.build`` :213-214, ``build_from_cfg`` :9-56).
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg and not (default_args and "type" in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
    if not isinstance(registry, Registry):
        raise TypeError(f"registry must be a Registry, but got {type(registry)}")
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type) or callable(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    try:
        return obj_cls(**args)
    except Exception as e:  # add the class name, plain TypeErrors do not carry it
        raise type(e)(f"{obj_cls.__name__}: {e}") from e


class Registry:
    def __init__(self, name, build_func=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return f"Registry(name={self._name}, items={sorted(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, names, force):
        if not (inspect.isclass(cls) or callable(cls)):
            raise TypeError(f"module must be a class or callable, but got {type(cls)}")
        names = [cls.__name__] if names is None else ([names] if isinstance(names, str) else names)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f"{n} is already registered in {self._name}")
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f"name must be None, a str or a sequence of str, got {type(name)}")
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls

        return deco
