#!/usr/bin/env python3
"""Turn gpurun_out/r04 (tools/prof_r04.sh) into the files kept under profiles/ (r04_*), and regenerate from them - and from
nothing else - the numbers quoted in the text: the section between `<!-- r04:begin -->` and `<!-- r04:end -->` of
profiles/README.md and of DESIGN.md section 6, and the header of profiles/r04_rocprofv3_pmc_summary.txt.  VERDICT r02 found
README / DESIGN / the summary header quoting numbers of an earlier collection; with this script a number is never typed."""
import collections
import csv
import glob
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = os.path.join(R, "gpurun_out", "r04")
P = os.path.join(R, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


# ---- bench lines ----------------------------------------------------------------------------------------------------
lines = {}
for f in sorted(glob.glob(os.path.join(b, "bench_*.json"))):
    name = os.path.basename(f)[6:-5]
    try:
        lines[name] = last_json(f)
    except Exception as e:
        print("skip", f, e)
head = lines["driverflags"]
open(os.path.join(P, "r04_bench_n1.json"), "w").write(json.dumps(head) + "\n")
with open(os.path.join(P, "r04_bench_lines.jsonl"), "w") as fo:
    for k, d in lines.items():
        d = dict(d)
        d["_run"] = k
        fo.write(json.dumps(d) + "\n")
order = ["driverflags", "default", "overlap", "grid64", "mfma1", "dot2", "c128", "c256", "c1024", "grid1024", "cfg5_256", "cfg5_256_streamed",
         "t512_auto", "t512_mfma1s", "t256_auto", "t256_mfma1s", "pocsag_d25", "multifm_d40", "multifm_d40_mfma1"]
what = {"driverflags": "cfg2, driver's flags", "default": "cfg2, defaults",
        "overlap": "cfg2, defaults, consecutive launches on two streams (MFM_F_OVERLAP; a launch's duration then includes its wait for workgroup slots)", "grid64": "cfg2 geometry, every channel on the 12.5 kHz raster",
        "mfma1": "cfg2, first-generation kernel", "dot2": "cfg2, v_dot2 kernel", "c128": "128 channels (configs[2] shard)",
        "c256": "256 channels", "c1024": "1024 channels on one GPU", "grid1024": "1024 channels on the 12.5 kHz raster",
        "cfg5_256": "configs[4] per-GPU share: 256 ch, D = 400, 512 taps (taps resident)",
        "cfg5_256_streamed": "the same, taps streamed from L2 (MFM_F_STREAM_TAPS, the round-1/2 form)",
        "t512_auto": "64 ch, D = 96, 512 taps (taps resident)", "t512_mfma1s": "the same, taps streamed",
        "t256_auto": "64 ch, D = 96, 256 taps (taps resident)", "t256_mfma1s": "the same, taps streamed", "pocsag_d25": "etc/pocsag_rtlsdr.json geometry: 64 ch, D = 25",
        "multifm_d40": "etc/multifm.json geometry: 64 ch, 1 MS/s, D = 40", "multifm_d40_mfma1": "the same, first-generation kernel"}
hdr = "run                kernel                     value(MSamp/s x ch)  ms/step  kernel_ms  min     median  p95     hbm_frac  compute_frac  issued  verified"
rows = [hdr]
table_md = ["| run | kernel | value (MSamp/s x ch) | ms per step | kernel ms (min / median / p95) | roofline.frac | int8 frac (issued) | verified |",
            "|---|---|---|---|---|---|---|---|"]
for k in order + [k for k in lines if k not in order]:
    if k not in lines:
        continue
    d = lines[k]
    r, c = d["roofline"], d["compute_roofline"]
    rows.append(f"{k:18s} {r['kernel']:26s} {d['value']:14.4g}      {d['ms_per_step']:.4f}   {r['kernel_ms']:.4f}   {r['kernel_ms_min']:.4f}  "
                f"{r['kernel_ms_median']:.4f}  {(r['kernel_ms_p95'] or float('nan')):.4f}  {r['frac']:.3f}     {c['frac']:.3f}         "
                f"{c.get('frac_issued', float('nan')):.3f}   {d.get('verified')}")
    table_md.append(f"| {what.get(k, k)} | {r['kernel'].replace('mfm_channel_kernel', 'kernel')} | {d['value'] / 1e6:.1f} M | {d['ms_per_step']:.4f} | "
                    f"{r['kernel_ms']:.4f} ({r['kernel_ms_min']:.4f} / {r['kernel_ms_median']:.4f} / {(r['kernel_ms_p95'] or float('nan')):.4f}) | {r['frac']:.3f} | "
                    f"{c['frac']:.3f} ({c.get('frac_issued', float('nan')):.3f}) | {d.get('verified')} |")
open(os.path.join(P, "r04_bench_table.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))

# ---- rocprofv3 kernel stats of the headline command -------------------------------------------------------------------
trace = {}
for tag in ("kstats", "kstats1024"):
    for f in glob.glob(os.path.join(b, tag, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, f"r04_rocprofv3_kernel_stats{'' if tag == 'kstats' else '_1024ch'}.csv"))
    tr = glob.glob(os.path.join(b, tag, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rr = [r for r in csv.DictReader(open(tr[0])) if "channel_kernel" in r["Kernel_Name"]]
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in sorted(rr, key=lambda r: int(r["Start_Timestamp"]))]
        trace[tag] = {"launches": len(du), "last20_us": sum(du[-20:]) / 20, "all_us": sum(du) / len(du), "name": rr[0]["Kernel_Name"]}
        print(tag, trace[tag])
        if tag == "kstats":
            open(os.path.join(P, "r04_kernel_duration_series.txt"), "w").write(
                "# launch durations (us) of the channel kernel in the rocprofv3 trace of `bench.py --gpus 1 --steps 20 --warmup 5`:\n"
                "# settle phase first, the last 25 launches are warm-up + timed region\n" + "\n".join("%.1f" % x for x in du) + "\n")


def stats_avg(path, needle):
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1000, int(r["Calls"]), r["Name"]
    return None


ks = stats_avg(os.path.join(P, "r04_rocprofv3_kernel_stats.csv"), "channel_kernel")
ks1024 = stats_avg(os.path.join(P, "r04_rocprofv3_kernel_stats_1024ch.csv"), "channel_kernel")

# ---- PMC passes -------------------------------------------------------------------------------------------------------
pm = {}
out = []
for p in ("p1", "p2", "p3"):
    fs = glob.glob(os.path.join(b, p, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    kn = ""
    for r in csv.DictReader(open(fs[0])):
        if "channel_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kn = r["Kernel_Name"]
    out.append(f"## pass {p}: {kn[:90]}")
    for k, v in sorted(acc.items()):
        m = sum(v[len(v) // 2:]) / len(v[len(v) // 2:])
        pm[k] = m
        out.append(f"{k:28s} launches={len(v):3d} mean={m:.6g}")
open(os.path.join(P, "r04_rocprofv3_pmc_raw.txt"), "w").write("\n".join(out) + "\n")

# ---- HBM traffic ------------------------------------------------------------------------------------------------------
res = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    fs = glob.glob(os.path.join(b, d, "**", "*counter_collection.csv"), recursive=True)
    rr = list(csv.DictReader(open(fs[0])))
    vals = [float(r["Counter_Value"]) for r in rr if "channel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals = vals[len(vals) // 2:]
    res[name] = sum(vals) / max(1, len(vals))
    shutil.copy(fs[0], os.path.join(P, f"r04_pmc_{name}.csv"))
alg = head["roofline"]["bytes_per_launch"]
tr = {"kernel": head["roofline"]["kernel"], "workload": head["config"]["workload"],
      "FETCH_SIZE_kb_per_launch": res["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": res["WRITE_SIZE"],
      "correction": "gfx950: FETCH_SIZE counts 16 B/lane streaming reads at half their bytes (MI355X_MICROARCH.md): x2",
      "hbm_bytes_per_launch": (res["FETCH_SIZE"] * 2 + res["WRITE_SIZE"]) * 1024, "algorithmic_bytes_per_launch": alg}
tr["ratio"] = tr["hbm_bytes_per_launch"] / alg
json.dump(tr, open(os.path.join(P, "r04_hbm_traffic.json"), "w"), indent=1)

# ---- 1024 channels: the same two counters ------------------------------------------------------------------------------
try:
    r1024 = {}
    for name, d in (("FETCH_SIZE", "fetch1024"), ("WRITE_SIZE", "write1024")):
        fs = glob.glob(os.path.join(b, d, "**", "*counter_collection.csv"), recursive=True)
        vv = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "channel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
        vv = vv[len(vv) // 2:]
        r1024[name] = sum(vv) / max(1, len(vv))
    alg1024 = lines["c1024"]["roofline"]["bytes_per_launch"]
    t1024 = {"workload": lines["c1024"]["config"]["workload"], "FETCH_SIZE_kb_per_launch": r1024["FETCH_SIZE"],
             "WRITE_SIZE_kb_per_launch": r1024["WRITE_SIZE"], "hbm_bytes_per_launch": (2 * r1024["FETCH_SIZE"] + r1024["WRITE_SIZE"]) * 1024,
             "algorithmic_bytes_per_launch": alg1024}
    t1024["ratio"] = t1024["hbm_bytes_per_launch"] / alg1024
    json.dump(t1024, open(os.path.join(P, "r04_hbm_traffic_1024ch.json"), "w"), indent=1)
    print("1024 channels:", t1024)
except Exception as e:
    t1024 = None
    print("no 1024-channel traffic:", e)

# ---- calibration of the two byte counters on kernels of known byte counts (tools/ubench_hbm.hip calib) ------------------------
cal = ["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on kernels that move a known number of bytes (tools/ubench_hbm.hip calib, tools/prof_r04.sh);",
       "# counter value in KB per launch (mean), the bytes the kernel really moves, and counter / bytes.",
       "#   pcm8<false> / pcm8<true>: the second-generation kernel's PCM store pattern (8 bytes per lane, 128 contiguous bytes per channel",
       "#   row and tile, 64 rows, 10 923 tiles = 89.5 MB), plain and with the non-temporal hint; wr: 16-byte streaming stores; rd: 16-byte",
       "#   streaming loads of 268 MB."]
known = {"pcm8<false>": 10923 * 64 * 128, "pcm8<true>": 10923 * 64 * 128, "rd": 268435456, "wr": 64 * 699072 * 2}
for name, d in (("FETCH_SIZE", "cal_fetch"), ("WRITE_SIZE", "cal_write")):
    fs = glob.glob(os.path.join(b, d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        m = sum(v) / len(v)
        kb = [x for x in known if k.startswith(x.split("<")[0]) and (("<" not in x) or x in k)]
        byt = known[kb[0]] if kb else None
        rel = (name == "FETCH_SIZE" and k.startswith("rd")) or (name == "WRITE_SIZE" and not k.startswith("rd"))
        cal.append(f"{name:11s} {k:14s} {m:12.0f} KB" + (f"   bytes moved {byt / 1024:.0f} KB   counter / bytes {m * 1024 / byt:.3f}" if byt and rel else ""))
open(os.path.join(P, "r04_counter_calibration.txt"), "w").write("\n".join(cal) + "\n")
print("\n".join(cal))

# ---- the PMC summary, every figure derived here -------------------------------------------------------------------------
simds = 1024.0
wave_cyc = pm.get("SQ_WAVE_CYCLES", float("nan"))
launch_cyc = pm.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0  # summed over the 8 XCDs
mfma_busy = pm.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
n_mfma, n_valu = pm.get("SQ_INSTS_MFMA", float("nan")), pm.get("SQ_INSTS_VALU", float("nan"))
other_valu = n_valu - n_mfma
mfma_frac = mfma_busy / (simds * launch_cyc)
valu_frac = other_valu * 4.0 / (simds * launch_cyc)  # upper bound: every other VALU instruction charged a full 4-cycle issue
valu_frac3 = other_valu * 3.0 / (simds * launch_cyc)  # round 2's convention (its 38 %): 3 cycles per instruction on average
valu_frac2 = other_valu * 2.0 / (simds * launch_cyc)  # lower bound: everything in the 2-cycle class
pairs = head["config"]["channels_per_gpu"] * (head["config"]["block_samples"] // 96)
summ = [
    "# rocprofv3 --pmc summary, round 4 (tools/prof_r04.sh: `bench.py --steps 8 --warmup 3 --settle-seconds 0.3`, cfg2: 64 ch, block 2^26);",
    "# mean per launch over the second half of the profiled launches.  GENERATED by tools/collect_r04.py from r04_rocprofv3_pmc_raw.txt -",
    "# every figure below is computed from the counters in this file, none is typed.",
    f"#   kernel                               {out[0][11:] if out else '?'}",
    f"#   launch length                        GRBM_GUI_ACTIVE / 8 XCDs = {launch_cyc:.4g} cycles",
    f"#   SQ_INSTS_VALU (incl. MFMA)           {n_valu:.4g}   SQ_INSTS_MFMA {n_mfma:.4g}   other VALU {other_valu:.4g} "
    f"= {other_valu * 64 / pairs:.1f} lane-instructions per (channel, output)",
    f"#   matrix pipe busy                     SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch) = {100 * mfma_frac:.1f} %",
    f"#   other VALU                           {100 * valu_frac2:.1f} / {100 * valu_frac3:.1f} / {100 * valu_frac:.1f} % at 2 / 3 / 4 cycles per instruction (fp32 mul/add/fma and "
    "32-bit add/logic issue in 2, the rest in ~4: profiles/r02_ubench_ops.txt; round 2 quoted the 3-cycle figure)",
    f"#   neither (3-cycle convention)         {100 * (1 - mfma_frac - valu_frac3):.1f} %   (round 2: 36 %)",
    f"#   MFMA time with a VALU instruction beside it   SQ_VALU_MFMA_COEXEC_CYCLES / MFMA_BUSY = "
    f"{100 * pm.get('SQ_VALU_MFMA_COEXEC_CYCLES', float('nan')) / mfma_busy:.0f} %",
    f"#   waves waiting (any reason)           SQ_WAIT_ANY / SQ_WAVE_CYCLES = {100 * pm.get('SQ_WAIT_ANY', float('nan')) / wave_cyc:.0f} %;"
    f" for an issue slot: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {100 * pm.get('SQ_WAIT_INST_ANY', float('nan')) / wave_cyc:.0f} %",
    f"#   LDS bank conflicts                   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = "
    f"{100 * pm.get('SQ_LDS_BANK_CONFLICT', float('nan')) / pm.get('SQ_LDS_IDX_ACTIVE', float('nan')):.0f} %",
    f"#   VMEM instructions                    {pm.get('SQ_INSTS_VMEM_RD', float('nan')):.4g} loads, {pm.get('SQ_INSTS_VMEM_WR', float('nan')):.4g} stores",
    f"#   HBM traffic (separate passes)        {tr['hbm_bytes_per_launch'] / 1e6:.1f} MB = {tr['ratio']:.3f} x algorithmic ({alg / 1e6:.1f} MB)",
    "#",
]
open(os.path.join(P, "r04_rocprofv3_pmc_summary.txt"), "w").write("\n".join(summ + out) + "\n")
print("\n".join(summ))

# ---- the issue model behind roofline.ceiling_frac (bench.py reads this file) -----------------------------------------------
# A SIMD issues one instruction stream: matrix instructions (16 cycles each at this shape) and other vector instructions do
# not overlap on gfx950 except for two in an MFMA's shadow.  If the SIMDs never waited, a launch would
# take (MFMA busy cycles + other VALU active cycles) / 1024 SIMDs at the sustained clock; what the kernel takes beyond that is
# latency it does not hide.  Cycles from the counters themselves: SQ_VALU_MFMA_BUSY_CYCLES, and SQ_ACTIVE_INST_VALU (x 4: the
# counter ticks per quad-cycle) less the MFMA share.
clock_ghz = launch_cyc / (head["roofline"]["kernel_ms"] * 1e6) if launch_cyc == launch_cyc else float("nan")
valu_active = pm.get("SQ_ACTIVE_INST_VALU", float("nan")) * 4.0
issue = {"source": "profiles/r04_rocprofv3_pmc_raw.txt (tools/prof_r04.sh, tools/collect_r04.py)",
         "workload": head["config"]["workload"], "kernel": head["roofline"]["kernel"],
         "mfma_instructions_per_launch": n_mfma, "other_valu_instructions_per_launch": other_valu,
         "mfma_busy_cycles_per_simd": mfma_busy / simds, "valu_cycles_per_simd_at_2": other_valu * 2.0 / simds,
         "valu_cycles_per_simd_at_3": other_valu * 3.0 / simds, "valu_cycles_per_simd_at_4": other_valu * 4.0 / simds,
         "launch_cycles": launch_cyc, "clock_ghz_in_profiled_run": clock_ghz,
         "busy_fraction_3cycle": mfma_frac + valu_frac3,
         "note": "ceiling = the launch with every SIMD cycle spent issuing (3 cycles per non-matrix vector instruction, the mix "
                 "measured in profiles/r02_ubench_ops.txt): time x busy_fraction"}
json.dump(issue, open(os.path.join(P, "r04_issue_model.json"), "w"), indent=1)
print("issue model:", issue)

# ---- generated text: profiles/README.md and DESIGN.md section 6 ----------------------------------------------------------
host = open(os.path.join(b, "host.txt")).read().split("\n") if os.path.exists(os.path.join(b, "host.txt")) else ["?", "?"]
cb = head.get("cpu_baseline", {})
gen = []
gen.append(f"Generated by `tools/collect_r04.py` from `gpurun_out/r04` (`tools/prof_r04.sh`, one box, one `gpurun` call); edit the script, not this text.")
gen.append("")
gen.append(f"* Headline (`r04_bench_n1.json`, the driver's command `python bench.py --gpus 1 --steps 20 --warmup 5`): "
           f"**{head['value'] / 1e6:.1f} M MSamp/s x channels**, `ms_per_step` {head['ms_per_step']:.4f}, kernel {head['roofline']['kernel_ms']:.4f} ms "
           f"(HIP events on {head['roofline'].get('timed_launches')} of {head['roofline'].get('launches')} timed launches), `roofline.frac` "
           f"**{head['roofline']['frac']:.3f}**, `verified` {head.get('verified')} ({head.get('verification', {}).get('outputs_checked')} outputs of the last "
           f"timed launch against the oracle), {head.get('rotators', {}).get('exact_channels')} of {head['config']['channels_per_gpu']} rotators exact.")
if ks:
    gen.append(f"* `rocprofv3 --kernel-trace --stats` of the same command (`r04_rocprofv3_kernel_stats.csv`): `{ks[2][:60]}` averages "
               f"**{ks[0]:.1f} us** over {ks[1]} launches (settle phase included); the last 20 launches of the trace average "
               f"{trace.get('kstats', {}).get('last20_us', float('nan')):.1f} us (`r04_kernel_duration_series.txt`).")
if ks1024:
    gen.append(f"* 1024 channels on one GPU (`r04_rocprofv3_kernel_stats_1024ch.csv`): {ks1024[0] / 1000:.3f} ms per launch in the trace, "
               f"{lines['c1024']['roofline']['kernel_ms']:.3f} ms by the engine's events in the un-profiled run.")
gen.append(f"* HBM traffic (`r04_hbm_traffic.json`; FETCH_SIZE and WRITE_SIZE each in its own `--pmc` pass, FETCH_SIZE doubled for gfx950): "
           f"{res['FETCH_SIZE']:.0f} KB and {res['WRITE_SIZE']:.0f} KB per launch -> **{tr['hbm_bytes_per_launch'] / 1e6:.1f} MB = "
           f"{tr['ratio']:.3f} x algorithmic** ({alg / 1e6:.1f} MB).")
gen.append(f"* SQ counters (`r04_rocprofv3_pmc_summary.txt`): matrix pipe busy {100 * mfma_frac:.1f} % of the launch's SIMD cycles, other VALU "
           f"{100 * valu_frac3:.1f} % (3 cycles per instruction, round 2's convention; {100 * valu_frac2:.1f} .. {100 * valu_frac:.1f} % at 2 .. 4), "
           f"neither {100 * (1 - mfma_frac - valu_frac3):.1f} % (round 2: 36 %); "
           f"{other_valu * 64 / pairs:.1f} lane-instructions per (channel, output).")
fp = head.get("fp32_iq_path")
if fp:
    gen.append(f"* Float path in the same run (`fp32_iq_path`: {fp['block_samples']}-sample blocks like the headline, settle phase first): "
               f"{fp['ms_per_block']:.4f} ms per block = {fp['achieved_tflops']:.1f} TFLOP/s = **{fp['frac']:.3f} of the fp32 matrix peak**, "
               f"{fp['time_vs_int16_path']:.2f} x the integer kernel's time.")
if cb:
    gen.append(f"* CPU baseline in the same run: {cb.get('value', 0):.0f} MSamp/s x channels on {cb.get('cores')} threads of a "
               f"{cb.get('host_cores')}-core host ({cb.get('host_cpu')}); one channel on one core: {cb.get('msamp_per_s_one_channel_one_core', 0):.0f} MSamp/s.")
gen.append("")
gen += table_md
gen_text = "\n".join(gen)
open(os.path.join(P, "r04_summary.md"), "w").write(gen_text + "\n")

# single values quoted in running text: <!--r04:KEY-->value<!--/r04-->
d40v3 = lines.get("multifm_d40", {}).get("roofline", {}).get("kernel_ms", float("nan"))
d40v1 = lines.get("multifm_d40_mfma1", {}).get("roofline", {}).get("kernel_ms", float("nan"))
vals = {"ms_step": f"{head['ms_per_step']:.4f}", "kernel_ms": f"{head['roofline']['kernel_ms']:.4f}", "frac": f"{head['roofline']['frac']:.3f}",
        "gap_us": f"{(head['ms_per_step'] - head['roofline']['kernel_ms']) * 1e3:.1f}", "d40_v3": f"{d40v3:.4f}", "d40_v1": f"{d40v1:.4f}",
        "d40_gain": f"−{100 * (1 - d40v3 / d40v1):.0f} %"}
for key, a, b_ in (("cfg5", "cfg5_256", "cfg5_256_streamed"), ("t512", "t512_auto", "t512_mfma1s"), ("t256", "t256_auto", "t256_mfma1s")):
    if a in lines and b_ in lines:
        ra, rb = lines[a]["roofline"]["kernel_ms"], lines[b_]["roofline"]["kernel_ms"]
        vals[key + "_res"], vals[key + "_str"], vals[key + "_gain"] = f"{ra:.4f}", f"{rb:.4f}", f"−{100 * (1 - ra / rb):.0f} %"
        vals[key + "_mfma"] = f"{lines[a]['compute_roofline']['frac']:.2f}"
# round 4's additions: block series, host-fed figures, overlap, issue ceiling, other geometries
bs = head.get("block_series", {}).get("series", [])
def _ser(blog, mode, key="frac"):
    for row in bs:
        if row.get("block_samples") == 1 << blog and isinstance(row.get(mode), dict) and key in row[mode]:
            return row[mode][key]
    return float("nan")
vals.update({"s20_one": f"{_ser(20, 'per_block_one_stream'):.3f}", "s20_co": f"{_ser(20, 'coalesced'):.3f}",
             "s22_one": f"{_ser(22, 'per_block_one_stream'):.3f}", "s22_co": f"{_ser(22, 'coalesced'):.3f}",
             "traffic_ratio": f"{tr['ratio']:.3f}", "mfma_pct": f"{100 * mfma_frac:.0f}", "valu_pct": f"{100 * valu_frac3:.0f}",
             "ceiling": f"{head['roofline']['frac'] / (mfma_frac + valu_frac3):.2f}"})
if "overlap" in lines:
    vals["ms_overlap"] = f"{lines['overlap']['ms_per_step']:.4f}"
    vals["ms_step"] = f"{lines['default']['ms_per_step']:.4f}" if "default" in lines else vals["ms_step"]
og = head.get("other_geometries", {})
if "configs3_pocsag_d25" in og and "kernel_ms" in og["configs3_pocsag_d25"]:
    vals["d25"] = f"{og['configs3_pocsag_d25']['kernel_ms']:.3f}"
if "configs4_int16_share" in og and "kernel_ms" in og["configs4_int16_share"]:
    vals["cfg5"] = f"{og['configs4_int16_share']['kernel_ms']:.3f}"
if head.get("fp32_iq_path"):
    vals["f32_frac"] = f"{head['fp32_iq_path']['frac']:.2f}"
st = ["| block | mode | launches | us per block | input GSamp/s | of the HBM roof |", "|---|---|---|---|---|---|"]
names = {"coalesced": "backlog gathered into launches of up to 2^26 samples, two streams", "per_block": "every block its own launch, two streams",
         "coalesced_one_stream": "gathered, one stream", "per_block_one_stream": "every block its own launch, one stream (rounds 1-3)"}
for row in bs:
    for mode in ("coalesced", "coalesced_one_stream", "per_block", "per_block_one_stream"):
        m = row.get(mode)
        if isinstance(m, dict) and "frac" in m:
            st.append(f"| 2^{row['block_samples'].bit_length() - 1} x {row['blocks']} | {names[mode]} | {m['launches']} | {m['us_per_block']:.2f} | "
                      f"{m['input_msamp_per_s'] / 1e3:.1f} | **{m['frac']:.3f}** |")
ee = head.get("end_to_end", {})
st.append("")
st.append(f"Host-fed (`end_to_end`, {ee.get('buffer_samples')}-sample buffers, {ee.get('buffers')} of them; PCIe both ways inside the figure):")
st.append("")
st.append("| mode | launches | us per buffer | input GSamp/s | H2D GB/s | D2H GB/s |")
st.append("|---|---|---|---|---|---|")
for mode, m in ee.items():
    if isinstance(m, dict) and "input_msamp_per_s" in m:
        st.append(f"| {mode} | {m['launches']} | {m['us_per_buffer']:.1f} | {m['input_msamp_per_s'] / 1e3:.2f} | {m['h2d_GBps']:.1f} | {m['d2h_GBps']:.1f} |")
cba = cb.get("all_cores") if cb else None
if cba:
    st.append("")
    st.append(f"CPU baseline on all cores (`cpu_baseline.all_cores`): {cba['value']:.0f} MSamp/s x channels on {cba['cores']} threads, {cba['channels']} channels.")
series_text = "\n".join(st)
open(os.path.join(P, "r04_block_series.md"), "w").write(series_text + "\n")

# the exchange table of DESIGN.md section 7
blk_mb = head["config"]["block_samples"] * 4 / 1e6
xt = ["| channels per GPU | kernel per block | needed per peer (int16 / 8-bit) | broadcast (≈ 153 GB/s per GPU) | all-gather on 7 links (≈ 940 GB/s at N = 8) |",
      "|---|---|---|---|---|"]
for key, lab in (("driverflags", "64"), ("c128", "128 (configs[2]: 1024 on 8 GPUs)"), ("c256", "256"), ("c1024", "1024")):
    if key not in lines:
        continue
    k = lines[key]["roofline"]["kernel_ms"]
    need = blk_mb / k  # MB per ms = GB/s
    f = lambda have, n: "hidden" if n <= have else f"{n / have:.1f} x short"
    xt.append(f"| {lab} | {k:.3f} ms | {need:.0f} / {need / 2:.0f} GB/s | {f(153.0, need)} | {f(940.0, need)} (8-bit: {f(940.0, need / 2)}) |")
xt_text = "\n".join(xt)

for path in (os.path.join(P, "README.md"), os.path.join(R, "DESIGN.md")):
    s = open(path).read()
    if "<!-- r04:begin -->" not in s:
        print("no r04 markers in", path)
        continue
    s = re.sub(r"<!-- r04:begin -->.*?<!-- r04:end -->", lambda m: "<!-- r04:begin -->\n" + gen_text + "\n<!-- r04:end -->", s, flags=re.S)
    s = re.sub(r"<!-- r04x:begin -->.*?<!-- r04x:end -->", "<!-- r04x:begin -->\n" + xt_text + "\n<!-- r04x:end -->", s, flags=re.S)
    s = re.sub(r"<!-- r04s:begin -->.*?<!-- r04s:end -->", lambda m: "<!-- r04s:begin -->\n" + series_text + "\n<!-- r04s:end -->", s, flags=re.S)
    for k, v in vals.items():
        s = re.sub(r"<!--r04:%s-->.*?<!--/r04-->" % k, "<!--r04:%s-->%s<!--/r04-->" % (k, v), s)
    open(path, "w").write(s)
    print("regenerated the r04 section of", path)
