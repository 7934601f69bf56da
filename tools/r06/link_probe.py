#!/usr/bin/env python3
"""Round 5: the host -> device link in the pieces the receiver's pool has (512 KiB sample_bufs, 64 bytes of header between
them), one command per piece against one strided command per run of adjacent pieces.  tools/r05/link_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

lib = ge.load_package().load_library()
for piece, gap, per in ((512 << 10, 0, 1), (512 << 10, 64, 1), (512 << 10, 64, 2), (512 << 10, 64, 4), (512 << 10, 64, 8), (512 << 10, 64, 16),
                        (512 << 10, 64, 32), (512 << 10, 64, 128), (16 << 10, 64, 1), (16 << 10, 64, 64), (16 << 10, 64, 512), (64 << 20, 0, 1)):
    for back in (0.0, 1.0 / 3.0):
        h, d = C.c_double(), C.c_double()
        rc = lib.mfm_link_probe_runs(0, piece, gap, per, 2 << 30, back, C.byref(h), C.byref(d))
        print(f"piece {piece >> 10:6d} KiB gap {gap:3d} pieces/command {per:4d} d2h share {back:.2f}: rc {rc} H2D {h.value:6.2f} GB/s D2H {d.value:6.2f} GB/s", flush=True)
