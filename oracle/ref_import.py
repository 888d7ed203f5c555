"""Import the *real* reference (read-only checkout at /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (never on the GPU box).  The reference's
hot-path modules import a few network / IO packages that are absent here and unused on the path
(`vilbert/file_utils.py:20-21` boto3; `utils/misc.py:16-19` colorama/termcolor/pyfiglet; `features_reader.py:6` lmdb;
`scripts/video_process/gen_instructions4train.py:7,21` argtyped; `pretrain.py:6` tensorboardX).  Empty stand-in modules
are registered for those names so that `vilbert.vilbert`, `vilbert.optimization`, `lily`, `utils.utils_init` import.
Nothing from the reference is copied; it is executed in place.
"""
from __future__ import annotations

import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name: str, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _stub("boto3")
    bc = _stub("botocore")
    bc.exceptions = _stub("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
    _stub("lmdb")
    _stub("tensorboardX", SummaryWriter=type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None}))
    _stub("colorama", init=lambda *a, **k: None, Fore=types.SimpleNamespace(), Style=types.SimpleNamespace())
    _stub("termcolor", cprint=lambda *a, **k: None, colored=lambda s, *a, **k: s)
    _stub("pyfiglet", figlet_format=lambda s, *a, **k: s)

    class Arguments:
        def __init_subclass__(cls, **kwargs):
            pass

        def __init__(self, *a, **k):
            pass

    _stub("argtyped", Arguments=Arguments)


def import_reference():
    """Returns a namespace with the reference modules on the hot path."""
    import os
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present (expected only in the build container)")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import vilbert.vilbert as rv
    import vilbert.optimization as ro
    import vilbert.vilbert_init as ri
    import lily as rl
    import utils.utils_init as ru
    return types.SimpleNamespace(vilbert=rv, optimization=ro, vilbert_init=ri, lily=rl, utils_init=ru)
