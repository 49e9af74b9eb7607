"""Import shim for the upstream reference (test infrastructure, runs ONLY in the build container).

This module is part of ``oracle/``: it is test infrastructure, never imported by the product
package ``saev_amd``.  It registers no-op stand-ins for Python packages the reference imports but
this image lacks (beartype, jaxtyping, tyro, orjson, wandb, torchvision), then puts
``/root/reference/src`` on ``sys.path`` so ``saev.nn.modeling``, ``saev.nn.objectives``,
``saev.utils.scheduling`` and ``saev.framework.train`` can be imported and *executed* to produce
golden vectors (``oracle/gen_golden.py``).  Nothing from the reference is copied; it cannot travel
to the GPU box, only the generated fixtures under ``tests/golden/`` do.
"""

import dataclasses
import enum
import importlib
import json
import os
import pathlib
import sys
import types

REFERENCE_ROOT = pathlib.Path(os.environ.get("SAEV_REFERENCE", "/root/reference"))


def available() -> bool:
    return (REFERENCE_ROOT / "src" / "saev" / "nn" / "modeling.py").exists()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Ann:
    def __class_getitem__(cls, item):
        return cls


class _Passthrough:
    def __class_getitem__(cls, item):
        return item


def _json_default(user_default):
    def inner(o):
        if dataclasses.is_dataclass(o) and not isinstance(o, type):
            return dataclasses.asdict(o)
        if isinstance(o, pathlib.PurePath):
            return str(o)
        if isinstance(o, enum.Enum):  # orjson serialises enums by value natively
            return o.value
        if user_default is not None:
            return user_default(o)
        raise TypeError(type(o))

    return inner


def _dumps(obj, default=None, option=None):
    indent = 2 if option and (option & 2) else None
    seps = (",", ": ") if indent else (",", ":")
    out = json.dumps(obj, default=_json_default(default), indent=indent, separators=seps,
                     sort_keys=bool(option and (option & 4)), ensure_ascii=False)
    if option and (option & 1):
        out += "\n"
    return out.encode()


_installed = False


def install():
    """Idempotently install the stand-ins and return the imported reference modules."""
    global _installed
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    if not _installed:
        if "beartype" not in sys.modules:
            _mod("beartype", beartype=lambda f=None, **kw: f if f is not None else (lambda g: g))
        if "jaxtyping" not in sys.modules:
            _mod(
                "jaxtyping",
                Float=_Ann, Int=_Ann, Int64=_Ann, Bool=_Ann, UInt8=_Ann, Shaped=_Ann,
                jaxtyped=lambda typechecker=None: (lambda f: f),
            )
        if "tyro" not in sys.modules:
            tyro = _mod("tyro")
            tyro.conf = _mod(
                "tyro.conf", Suppress=_Passthrough, EnumChoicesFromValues=_Passthrough,
                arg=lambda **kw: None,
            )
            tyro.extras = _mod("tyro.extras")
        if "orjson" not in sys.modules:
            _mod(
                "orjson", OPT_APPEND_NEWLINE=1, OPT_INDENT_2=2, OPT_SORT_KEYS=4,
                dumps=_dumps, loads=json.loads,
            )
        if "tomllib" not in sys.modules:
            import tomli

            sys.modules["tomllib"] = tomli
        import multiprocessing.queues as _mq

        if not hasattr(_mq.Queue, "__class_getitem__"):
            _mq.Queue.__class_getitem__ = classmethod(lambda cls, item: cls)
        import typing

        import typing_extensions

        for n in ("assert_never", "Self"):
            if not hasattr(typing, n):
                setattr(typing, n, getattr(typing_extensions, n))
        if "wandb" not in sys.modules:

            class _Run:
                id = "fake0001"
                summary = {}

                def log(self, *a, **k):
                    pass

                def finish(self):
                    pass

            _mod("wandb", init=lambda **kw: _Run(), Settings=lambda **kw: None,
                 Table=lambda **kw: None)
        if "torchvision" not in sys.modules:
            tv = _mod("torchvision")
            tv.datasets = _mod("torchvision.datasets", ImageFolder=type("ImageFolder", (), {}))
        sys.path.insert(0, str(REFERENCE_ROOT / "src"))
        import saev  # noqa: F401  (the reference package)

        pkg = types.ModuleType("saev.data")
        pkg.__path__ = [str(REFERENCE_ROOT / "src" / "saev" / "data")]
        sys.modules["saev.data"] = pkg
        saev.data = pkg
        shards = importlib.import_module("saev.data.shards")
        shuffled = importlib.import_module("saev.data.shuffled")
        pkg.ShuffledConfig = shuffled.Config
        pkg.ShuffledDataLoader = shuffled.DataLoader
        pkg.Metadata = shards.Metadata
        _installed = True

    from saev.nn import modeling, objectives
    import saev.framework.train as train
    import saev.utils.scheduling as scheduling

    return types.SimpleNamespace(
        modeling=modeling, objectives=objectives, train=train, scheduling=scheduling,
        data=sys.modules["saev.data"],
    )
