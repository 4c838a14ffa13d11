"""fba_pomdp_amd -- MI355X-native BA-POMCP engine (belief update + POUCT/RBAPOUCT search).

The product is libfba_hip.so (hand-written HIP for gfx950 behind the C-ABI of include/fba_hip.h);
this package is the thin ctypes binding used by the tests and bench.py.
"""
from . import _native
from ._native import (BELIEF_IMPORTANCE, BELIEF_REJECTION, MODEL_BA_FACTORED, MODEL_BA_TABLE, MODEL_POMDP,
                      build, build_cli, load)
from .engine import Engine, FbaError

__all__ = ["Engine", "FbaError", "build", "build_cli", "load", "_native", "MODEL_POMDP", "MODEL_BA_TABLE",
           "MODEL_BA_FACTORED", "BELIEF_REJECTION", "BELIEF_IMPORTANCE"]
