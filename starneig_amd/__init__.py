"""MI355X-native Hessenberg / Schur hot path behind StarNEig's C interface.

The product is ``libstarneig_amd.so`` (HIP, gfx950); this package only binds it.
"""
from . import lib  # noqa: F401
from .lib import *  # noqa: F401,F403
