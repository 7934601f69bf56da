#!/usr/bin/env python3
"""A/B of library variants on ONE box, with error bars and a stated rule.

    python tools/exp/ab.py [--reps 6] [--bench-args "..."] [--metric kernel_ms|cycles|ms_per_step] base=<lib.so> cand=<lib.so> ...

Every repetition runs every variant once, in rotating order (variant i first in repetition i, so that no variant always runs on a
cold or a warm board), each as `bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20
--settle-seconds 0.5 <bench args>` with MFM_LIB pointing at the variant's library.  Per variant: mean, standard deviation
(n - 1), min, max of the metric over the repetitions.  Against the FIRST variant (the baseline) a candidate is

    kept      if its mean is lower by more than 2 pooled standard deviations of the two means' difference
              ( |d| > 2 * sqrt(s_a^2 / n + s_b^2 / n) ),
    worse     if it is higher by more than that,
    neutral   otherwise - not evidence, whatever the sign.

`<lib.so>` may also be `flags:<bench flags>` - the shipped library with extra bench.py flags (e.g. `flags:--kernel slice64`).
The table goes to stdout and, with --out, to a file (profiles/r06_ab_*.txt are such files)."""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_once(spec, bench_args, metric):
    env = dict(os.environ)
    extra = []
    if spec.startswith("flags:"):
        extra = spec[len("flags:"):].split()
    elif spec:
        env["MFM_LIB"] = os.path.abspath(spec)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-fp32", "--no-chain", "--no-series", "--steps", "200",
           "--warmup", "20", "--settle-seconds", "0.5"] + bench_args + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if not lines:
        raise SystemExit("bench.py gave no line for %r:\n%s" % (spec, (r.stdout + r.stderr)[-2000:]))
    d = json.loads(lines[-1])
    roof = d["roofline"]
    clocks = roof.get("clocks") or {}
    val = {"kernel_ms": roof["kernel_ms"] * 1e3, "cycles": clocks.get("shader_ticks_median"), "ms_per_step": d["ms_per_step"] * 1e3}[metric]
    return float(val), bool(d.get("verified")), roof.get("kernel"), clocks.get("sclk_mhz_effective")


def stats(xs):
    n = len(xs)
    m = sum(xs) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in xs) / (n - 1)) if n > 1 else float("nan")
    return m, sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--bench-args", default="")
    ap.add_argument("--metric", choices=["kernel_ms", "cycles", "ms_per_step"], default="kernel_ms")
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+", help="name=<library or flags:...>; the first one is the baseline")
    a = ap.parse_args()
    if a.reps < 2:
        raise SystemExit("at least two repetitions")
    names, specs = [], []
    for v in a.variants:
        n, _, s = v.partition("=")
        names.append(n)
        specs.append(s)
    got = {n: [] for n in names}
    info = {}
    t0 = time.time()
    for rep in range(a.reps):
        order = list(range(len(names)))
        order = order[rep % len(order):] + order[:rep % len(order)]
        for i in order:
            val, ok, kern, mhz = run_once(specs[i], a.bench_args.split(), a.metric)
            got[names[i]].append(val)
            info[names[i]] = (ok, kern)
            print("rep %d %-12s %10.2f  verified %s  %s MHz" % (rep, names[i], val, ok, "%.0f" % mhz if mhz else "?"), flush=True)
    unit = {"kernel_ms": "us per launch (HIP events)", "cycles": "shader cycles per launch (median of the kernel's own stamps)",
            "ms_per_step": "us per step"}[a.metric]
    out = ["# A/B on one box, %d alternating repetitions per variant, %s; bench args: %s; %.0f s" % (a.reps, unit, a.bench_args or "(defaults)", time.time() - t0),
           "# rule: kept / worse when the means differ by more than 2 standard errors of their difference, else neutral (tools/exp/ab.py)",
           "%-14s %10s %8s %10s %10s  %-9s %s" % ("variant", "mean", "sd", "min", "max", "verdict", "kernel")]
    bm, bs = stats(got[names[0]])
    for n in names:
        m, sd = stats(got[n])
        if n == names[0]:
            verdict = "baseline"
        else:
            se = math.sqrt(sd * sd / a.reps + bs * bs / a.reps)
            d = m - bm
            verdict = "neutral" if abs(d) <= 2.0 * se else ("kept" if d < 0 else "worse")
            verdict += " (%+.2f %%, %.1f se)" % (100.0 * d / bm, abs(d) / se if se > 0 else float("inf"))
        out.append("%-14s %10.2f %8.2f %10.2f %10.2f  %-9s %s%s" % (n, m, sd, min(got[n]), max(got[n]), verdict, info[n][1],
                                                                 "" if info[n][0] else "  NOT VERIFIED"))
    text = "\n".join(out)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
