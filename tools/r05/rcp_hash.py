#!/usr/bin/env python3
"""Print mfm_devtest_rcp_table() of device 0: the hash of v_rcp_f32's table over all 2^23 significands (what
MFM_RCP_TABLE_HASH_GFX950 in include/multifm_hip.h must be on an MI355X), the ulp counts, and the 2^28-quotient sweep."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
lib = pkg.load_library()
h, bad, tried = C.c_uint64(), C.c_uint64(), C.c_uint64()
counts = (C.c_uint64 * 4)()
rc = lib.mfm_devtest_rcp_table(0, C.byref(h), counts, C.byref(bad), C.byref(tried))
print("rc %d hash 0x%016x one-ulp-low %d exact %d one-ulp-high %d other %d | sweep: %d wrong of %d" % (
    rc, h.value, counts[0], counts[1], counts[2], counts[3], bad.value, tried.value))
