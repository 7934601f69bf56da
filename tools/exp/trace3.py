#!/usr/bin/env python3
"""Per-phase timeline of the v3 kernel from the stamped scratch build (tools/exp/k3_trace.hip)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MFM_LIB"] = os.path.join(ROOT, "tools", "exp", "libexp_trace.so")
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch")
block = 1 << 26
eng = pkg.Engine(fs, decim, block, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
for o, g in zip(offs, gains):
    eng.add_channel(int(o), taps, float(g))
eng.commit()
for _ in range(40):
    eng.acquire_input(); eng.submit(block, wait_producer=False)
eng.sync()
buf = np.zeros(16 * 8 * 40 * 8, np.uint64)
assert eng.lib.mfm_v3_trace_read(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
t = buf.reshape(16, 8, 40, 8).astype(np.int64)
names = ["loads issued..matrix phase+combine", "epilogue", "stage_store", "barrier wait", "(next tile start)"]
# phases: 0->1 matrix, 1->2 epilogue, 2->3 stage store, 3->4 barrier, 4->next 0: loop overhead
ok = t[:, :, 2:20, :]
d01 = ok[..., 1] - ok[..., 0]; d12 = ok[..., 2] - ok[..., 1]; d23 = ok[..., 3] - ok[..., 2]; d34 = ok[..., 4] - ok[..., 3]
per = ok[:, :, 1:, 0] - ok[:, :, :-1, 0]
for nm, d in [("matrix phase", d01), ("epilogue", d12), ("stage_store", d23), ("barrier wait", d34), ("tile period", per)]:
    print(f"{nm:14s} mean {d.mean():8.0f}  min {d.min():6d}  p10 {np.percentile(d,10):7.0f} median {np.median(d):7.0f} p90 {np.percentile(d,90):7.0f} max {d.max():6d}")
print("per wave mean matrix phase:", d01.mean(axis=(0, 2)).round())
print("per wave mean epilogue:", d12.mean(axis=(0, 2)).round())
print("per wave mean barrier wait:", d34.mean(axis=(0, 2)).round())
print("wg0 wave0 tiles 2..8 stamps rel:", (t[0,0,2:9,:5]-t[0,0,2,0]).tolist())
