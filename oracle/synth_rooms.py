"""Moved to ``tools/synth_rooms.py`` (synthetic INPUT generation is not a restatement of the reference, and
``bench.py``'s measured legs must not import ``oracle/``); this shim keeps the old import path for the tests."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from tools.synth_rooms import *  # noqa: F401,F403,E402
from tools.synth_rooms import W, H  # noqa: F401,E402
