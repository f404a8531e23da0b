"""tests' name for oracle/core_ref.py (the ctypes face of oracle/_ref/libcore_ref.so)."""
from oracle.core_ref import *  # noqa: F401,F403
from oracle.core_ref import available, cartesian, destagger, lib  # noqa: F401
