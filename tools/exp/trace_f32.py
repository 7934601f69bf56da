"""timeline of two workgroups of the persistent float kernel (tools/exp/f32_trace.hip)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MFM_LIB"] = os.path.join(ROOT, "tools", "exp", "libexp_ftrace.so")
import torch
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
blk = 1 << 24
base = pkg.synth.synth_iq(1 << 20, fs, offs[::8][:8], seed=7)
iq = np.tile(base, (blk // base.shape[0] + 1, 1))[:blk]
d = torch.from_numpy(iq.astype(np.float32).reshape(-1)).cuda()
eng = pkg.F32Engine(fs, decim, blk, device=0)
for o, g in zip(offs, gains):
    eng.add_channel(int(o), taps, float(g))
eng.commit()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.process_device(d.data_ptr(), blk, stream=st)
torch.cuda.synchronize()
lib = pkg.load_library()
out = np.zeros(2 * 8 * 64 * 8, np.uint64)
assert lib.mfm_f32_debug_trace(out.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
t = out.reshape(2, 8, 64, 8).astype(np.int64)
t0 = t[t > 0].min()
for wg in range(2):
    for wave in (0, 3, 7):
        print(f"wg {wg} wave {wave}: per chunk  start | load-issue  phase  store  barrier  epilogue | total")
        for it in range(26):
            r = t[wg, wave, it]
            if r[0] == 0:
                break
            ep = (r[5] - r[4]) if r[5] > 0 else 0
            end = r[5] if r[5] > 0 else r[4]
            print(f"   it {it:2d}: {r[0]-t0:8d} | {r[1]-r[0]:6d} {r[2]-r[1]:7d} {r[3]-r[2]:6d} {r[4]-r[3]:7d} {ep:7d} | {end-r[0]:7d}")

d_keep = d
sp = np.zeros(2048, np.uint64)
assert lib.mfm_f32_debug_span(sp.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
sp = sp.reshape(1024, 2).astype(np.int64)
sp = sp[sp[:, 0] > 0]
st, en = sp[:, 0] - sp[:, 0].min(), sp[:, 1] - sp[:, 0].min()
print("workgroups", len(sp), "start min/median/max", st.min(), int(np.median(st)), st.max(), "end min/median/max", en.min(), int(np.median(en)), en.max())
dur = en - st
print("duration min/median/max", dur.min(), int(np.median(dur)), dur.max())
