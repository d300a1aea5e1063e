"""A small reader for the `Class.field = literal` files under the reference's configs/ (gin-config is
not a dependency here).  `@configurable` makes a dataclass fill constructor arguments it was not
given from the bindings parsed so far, so `parse_config_file(p); CartpoleDynamicsConfig()` works the
way the reference's entry points use gin (scripts/test_vhjb_policy.py:51-55).

Only literal bindings are supported (numbers, lists, booleans, strings) -- that is all the seven
reference files contain.  Macros, references (@x, %x), scopes and imports raise ValueError.
"""
from __future__ import annotations

import ast
import dataclasses
import functools

_BINDINGS: dict[str, dict[str, object]] = {}
_REGISTRY: dict[str, type] = {}


def clear_config():
    _BINDINGS.clear()


def parse_config(text: str):
    logical, buf, depth = [], "", 0
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].rstrip()
        if not line.strip() and depth == 0:
            continue
        buf += line
        depth = buf.count("[") + buf.count("(") + buf.count("{") - buf.count("]") - buf.count(")") - buf.count("}")
        if depth <= 0:
            logical.append(buf)
            buf, depth = "", 0
    if buf.strip():
        raise ValueError(f"unterminated binding: {buf!r}")
    for stmt in logical:
        if "=" not in stmt:
            raise ValueError(f"unsupported gin statement: {stmt!r}")
        key, val = stmt.split("=", 1)
        key, val = key.strip(), val.strip()
        if "." not in key or "/" in key or val[:1] in "@%":
            raise ValueError(f"unsupported gin binding: {stmt!r}")
        cls, field = key.rsplit(".", 1)
        _BINDINGS.setdefault(cls.split(".")[-1], {})[field] = ast.literal_eval(val)


def parse_config_file(path: str):
    with open(path) as f:
        parse_config(f.read())


def configurable(cls):
    """Class decorator: unspecified dataclass fields are taken from parsed bindings."""
    name = cls.__name__
    _REGISTRY[name] = cls
    orig_init = cls.__init__
    fields = [f.name for f in dataclasses.fields(cls)] if dataclasses.is_dataclass(cls) else []

    @functools.wraps(orig_init)
    def __init__(self, *args, **kwargs):
        bound = _BINDINGS.get(name, {})
        given = set(fields[: len(args)]) | set(kwargs)
        for k, v in bound.items():
            if k not in given:
                kwargs[k] = v
        orig_init(self, *args, **kwargs)

    cls.__init__ = __init__
    return cls
