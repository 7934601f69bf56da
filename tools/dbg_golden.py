import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
g = np.load(os.path.join(ROOT, "tests/golden/path_oracle.npz"))
eng = pkg.Engine(int(g["fs"]), int(g["decim"]), 1 << 15, device=0)
for c in range(3):
    eng.add_channel_q14(g["cre"][c], g["cim"][c], g["incr"][c], want_iq=True)
eng.commit()
print(eng.stats())
pcm, q = eng.run(g["iq"], 1 << 15)
bad = np.argwhere(pcm != g["pcm"])
print("pcm mismatches", len(bad), bad[:20].tolist())
badq = np.argwhere(q != g["filt_iq"])
print("iq mismatches", len(badq), badq[:20].tolist())
if len(bad):
    c, n = bad[0]
    print("hip", pcm[c, max(0,n-2):n+5], "ref", g["pcm"][c, max(0,n-2):n+5])
ref = g["filt_iq"]
for c in range(3):
    b = np.argwhere((q[c] != ref[c]).any(axis=1)).reshape(-1)
    print("chan", c, "bad outputs", b[:40].tolist())
c = 0
print("hip iq 75..95", q[c, 75:95].tolist())
print("ref iq 75..95", ref[c, 75:95].tolist())
# is the wrong data a shifted copy of the right data?
for sh in range(-40, 41):
    if sh and 77 + sh >= 0 and np.array_equal(q[c, 77:93], ref[c, 77 + sh:93 + sh]):
        print("hip[77:93] == ref shifted by", sh)
