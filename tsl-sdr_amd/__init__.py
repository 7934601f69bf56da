"""MI355X-native multifm channel engine (drop-in for pvachon/tsl-sdr's multifm hot path).

The product is libmultifm_hip.so (csrc/, behind include/multifm_hip.h) and the C host in host/.
This Python package only carries the ctypes mirror of that C ABI and the synthetic-input helpers
that tests/ and bench.py share.  The directory name has a hyphen, so load it with
`__graft_entry__.load_package()` (importlib), which registers it as `tsl_sdr_amd`.
"""
from . import binding, dist, synth  # noqa: F401
from .binding import Engine, F32Engine, Flex, Group, MfmError, MuellerMuller, Pocsag, Resampler, load_library  # noqa: F401
