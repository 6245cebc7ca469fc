"""quisk_amd -- MI355X (gfx950) receive DSP for Quisk / WDSP.

The product is the C-ABI shared library quisk_amd/lib/libquiskhip.so (include/quiskhip.h);
this package is the thin Python host side that mirrors the reference's own Python binding
(quisk_wdsp.py) on top of it.  There is no CPU fallback: every compute call needs a HIP device.
"""
from .lib import load, QuiskHipError          # noqa: F401
from .rxa import RxaEngine, AudioFormat       # noqa: F401
from .fir import FirBank, HalfBandCascade, RationalFir, hb45_taps           # noqa: F401
from .pan import Panadapter, Bandscope, waterfall_rows                   # noqa: F401
from .qrx import QuiskRxBank, QuiskProcessBank, QuiskAgc, NoiseBlanker                  # noqa: F401
from .analyzer import AnalyzerBank, WdspDisplay       # noqa: F401
from . import ingest                          # noqa: F401
from . import quiskapi                        # noqa: F401
from .ingest import IqFormat                  # noqa: F401

__all__ = ["load", "QuiskHipError", "RxaEngine", "AudioFormat", "FirBank", "HalfBandCascade", "RationalFir", "hb45_taps", "Panadapter", "Bandscope", "waterfall_rows", "QuiskRxBank", "QuiskProcessBank", "QuiskAgc", "NoiseBlanker", "AnalyzerBank", "WdspDisplay", "ingest", "IqFormat", "quiskapi"]
