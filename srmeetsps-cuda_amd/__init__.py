"""srmeetsps-cuda_amd -- MI355X (gfx950) implementation of the SRPS alternating-optimisation hot
path of nihalsid/SRmeetsPS-CUDA behind the reference's SRPS / DataHandler surface.

The directory name contains a hyphen (it is fixed by the project layout); import it with
``importlib.import_module("srmeetsps-cuda_amd")``.
"""
from ._lib import SRPSError, build, load, declared_symbols, LIB_PATH  # noqa: F401
from .api import (Context, DataHandler, Preferences, SRPS, TorchCollectives, alternating_loop, shard_range)  # noqa: F401
from . import synth  # noqa: F401


def last_error() -> str:
    """text of the calling thread's last failure or fallback notice (srps_last_error)"""
    return load().srps_last_error().decode(errors="replace")

from . import host  # noqa: F401
