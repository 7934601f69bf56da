#!/usr/bin/env python3
"""Round 5: bench.py's end_to_end object alone (host-fed figures, the link yardstick, paced-feed latency).  tools/r05/e2e.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
out = bench.end_to_end(pkg, fs, decim, taps, offs, gains)
for k, v in out.items():
    print(k, json.dumps(v))
