"""`model` package shim: the reference's train.py / test.py do `from model import *`; this makes that
import resolve to the MI355X-native implementation."""
from pesr_amd.model import *  # noqa: F401,F403
from pesr_amd.model import __all__  # noqa: F401
