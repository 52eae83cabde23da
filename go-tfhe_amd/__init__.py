"""go-tfhe_amd: MI355X (gfx950) gate-bootstrapping engine behind go-tfhe's gates.*/evaluator.* seams.

The compute lives in lib/libtfhe_hip.so (hand-written HIP, C ABI in include/tfhe_hip.h); this
package is the thin Python host side used by the tests and bench.py.  There is no CPU
fallback: importing works anywhere (so the CPU test tier can check the ABI), but creating a
context without a GPU, or with the library missing, raises.
"""
from . import params  # noqa: F401
from .params import (Params, Security80Bit, Security110Bit, Security128Bit, SecurityUint1, SecurityUint2,  # noqa: F401
                     SecurityUint3, SecurityUint4, SecurityUint5, SecurityUint6, SecurityUint7, SecurityUint8)
from ._binding import (Context, PinnedArray, TfheError, OPS, library_path, load_library, exported_symbols,  # noqa: F401
                       declared_symbols)
from .cloudkey import CloudKey, CloudKeySet  # noqa: F401
from . import gates, evaluator, lut, trgsw, trlwe  # noqa: F401
