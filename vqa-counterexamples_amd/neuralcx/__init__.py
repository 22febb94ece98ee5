"""neuralcx: host side of the MI355X-native NeuralCX hot path (HIP library + ctypes binding)."""
from . import _lib  # noqa: F401
from ._lib import NcxError, version  # noqa: F401
