"""The library's tuning table (include/sefd.h `sefd_tuning_*`, csrc/tuning.h): tile thresholds, ring depths, lane placement, ... of the planner and
the launchers.  ONE process-wide table - filled once from the environment variable SEFD_TUNING="KNOB=value,KNOB=value" and by the calls below -
instead of one environment variable per knob: a plan is a function of its configuration and of this table when it is built.

    from sefd_amd import tuning
    tuning.set("CG256_MINM", 64)          # any later Plan(...) sees it
    with tuning.scope(BN_FUSE=2): ...     # set for the block, restored afterwards
    tuning.unset("CG256_MINM"); tuning.clear()

Knobs and defaults: INTEGRATION.md section 6.  Nothing here changes results beyond floating-point summation order."""
import contextlib

from . import _lib


def _L():
    return _lib.lib()


def set(knob, value):          # noqa: A001 (module-level API: tuning.set)
    _L().sefd_tuning_set(str(knob).encode(), None if value is None else str(value).encode())


def unset(*knobs):
    for k in knobs:
        _L().sefd_tuning_set(str(k).encode(), None)


def get(knob):
    v = _L().sefd_tuning_get(str(knob).encode())
    return None if v is None else v.decode()


def clear():
    _L().sefd_tuning_clear()


@contextlib.contextmanager
def scope(**knobs):
    old = {k: get(k) for k in knobs}
    try:
        for k, v in knobs.items():
            set(k, v)
        yield
    finally:
        for k, v in old.items():
            set(k, v)
