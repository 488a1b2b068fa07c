"""Python-file configs with ``_base_`` inheritance, attribute access and ``--options`` overrides.

Dependency-free re-implementation (no addict / yapf) of the behaviour of ponder/utils/config.py
that the pre-training entry point relies on: ``Config.fromfile`` (:334-339) executing the file as
a module (:203-215), recursive ``_base_`` merge with ``_delete_`` (:241-271,280-331), ``ConfigDict``
attribute access (:33-48), ``merge_from_dict`` with dotted keys (:551-597) and ``DictAction``
(:600-694).  The reference's config files load unchanged.
"""
import ast
import copy
import os
import runpy
import types
from argparse import Action

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"
RESERVED_KEYS = ("filename", "text", "pretty_text")


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(x) for x in v]
        if isinstance(v, tuple):
            return tuple(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v
        return plain(self)

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _merge_a_into_b(a, b, allow_list_keys=False):
    b = copy.deepcopy(b)
    for k, v in a.items():
        if allow_list_keys and k.isdigit() and isinstance(b, list):
            k = int(k)
            if len(b) <= k:
                raise KeyError(f"Index {k} exceeds the length of list {b}")
            b[k] = _merge_a_into_b(v, b[k], allow_list_keys) if isinstance(v, dict) else v
        elif isinstance(v, dict):
            if k in b and not v.get(DELETE_KEY, False):
                allowed = (dict, list) if allow_list_keys else dict
                if not isinstance(b[k], allowed):
                    raise TypeError(
                        f"{k}={v} in child config cannot inherit from base because {k} is a dict "
                        f"in the child config but is of type {type(b[k])} in base config. You may "
                        f"set `{DELETE_KEY}=True` to ignore the base config.")
                b[k] = _merge_a_into_b(v, b[k], allow_list_keys)
            else:
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
                b[k] = copy.deepcopy(v)
        else:
            b[k] = v
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(f"config file {filename!r} does not exist")
    if not filename.endswith(".py"):
        raise IOError("Only py type configs are supported")
    with open(filename, "r", encoding="utf-8") as f:
        text = f.read()
    try:
        ast.parse(text)
    except SyntaxError as e:
        raise SyntaxError(f"There are syntax errors in config file {filename}: {e}")
    ns = runpy.run_path(filename)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType))}
    if BASE_KEY in cfg:
        bases = cfg.pop(BASE_KEY)
        bases = bases if isinstance(bases, (list, tuple)) else [bases]
        base_cfg, base_text = {}, []
        for b in bases:
            d, t = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = base_cfg.keys() & d.keys()
            if dup:
                raise KeyError(f"Duplicate key is not allowed among bases: {sorted(dup)}")
            base_cfg.update(d)
            base_text.append(t)
        cfg = _merge_a_into_b(cfg, base_cfg)
        text = "\n".join(base_text + [text])
    return cfg, text


class Config:
    def __init__(self, cfg_dict=None, cfg_text=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        for key in cfg_dict:
            if key in RESERVED_KEYS:
                raise KeyError(f"{key} is reserved for config file")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        if cfg_text is None and filename:
            with open(filename, "r") as f:
                cfg_text = f.read()
        object.__setattr__(self, "_text", cfg_text or "")

    @staticmethod
    def fromfile(filename, use_predefined_variables=True, import_custom_modules=True):
        cfg_dict, cfg_text = _file2dict(filename)
        return Config(cfg_dict, cfg_text=cfg_text, filename=filename)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    @property
    def pretty_text(self):
        def fmt(v, indent):
            pad = " " * indent
            if isinstance(v, dict):
                if not v:
                    return "dict()"
                rows = [f"{pad}    {k}={fmt(x, indent + 4)}," if str(k).isidentifier()
                        else f"{pad}    **{{{k!r}: {fmt(x, indent + 4)}}}," for k, x in v.items()]
                return "dict(\n" + "\n".join(rows) + f"\n{pad})"
            if isinstance(v, (list, tuple)) and any(isinstance(x, (dict, list, tuple)) for x in v):
                o, c = ("[", "]") if isinstance(v, list) else ("(", ")")
                rows = [f"{pad}    {fmt(x, indent + 4)}," for x in v]
                return o + "\n" + "\n".join(rows) + f"\n{pad}" + c
            return repr(v)
        return "\n".join(f"{k} = {fmt(v, 0)}" for k, v in self._cfg_dict.to_dict().items()) + "\n"

    def dump(self, file=None):
        if file is None:
            return self.pretty_text
        with open(file, "w") as f:
            f.write(self.pretty_text)

    def merge_from_dict(self, options, allow_list_keys=True):
        option_cfg_dict = {}
        for full_key, v in options.items():
            d = option_cfg_dict
            key_list = full_key.split(".")
            for subkey in key_list[:-1]:
                d = d.setdefault(subkey, {})
            d[key_list[-1]] = v
        merged = _merge_a_into_b(option_cfg_dict, self._cfg_dict.to_dict(),
                                 allow_list_keys=allow_list_keys)
        object.__setattr__(self, "_cfg_dict", ConfigDict(merged))

    def __repr__(self):
        return f"Config (path: {self.filename}): {self._cfg_dict!r}"

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __getstate__(self):
        return (self._cfg_dict, self._filename, self._text)

    def __setstate__(self, state):
        for k, v in zip(("_cfg_dict", "_filename", "_text"), state):
            object.__setattr__(self, k, v)


class DictAction(Action):
    """argparse action: ``--options a.b=1 c=[1,2] d=(x,y) e=true``."""

    @staticmethod
    def _scalar(val):
        for cast in (int, float):
            try:
                return cast(val)
            except ValueError:
                pass
        if val.lower() in ("true", "false"):
            return val.lower() == "true"
        if val == "None":
            return None
        return val

    @classmethod
    def _parse(cls, val):
        val = val.strip("'\" ").replace(" ", "")
        is_tuple = val.startswith("(") and val.endswith(")")
        if is_tuple or (val.startswith("[") and val.endswith("]")):
            val = val[1:-1]
        elif "," not in val:
            return cls._scalar(val)
        items, depth, cur = [], 0, ""
        for ch in val:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                items.append(cur)
                cur = ""
            else:
                cur += ch
        if cur:
            items.append(cur)
        out = [cls._parse(x) for x in items]
        return tuple(out) if is_tuple else out

    def __call__(self, parser, namespace, values, option_string=None):
        options = {}
        for kv in values:
            key, val = kv.split("=", maxsplit=1)
            options[key] = self._parse(val)
        setattr(namespace, self.dest, options)
