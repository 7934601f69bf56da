#!/usr/bin/env python3
"""Turn gpurun_out/r06e (tools/prof_r06.sh) - and gpurun_out/r06f (tools/prof_r06_final.sh: the driver's command once more, run
with the issue model of the first call in place) - into the files kept under profiles/ (r06_*).  Every number in
profiles/r06_summary.md, r06_block_series.md, r06_exchange_table.md and r06_values.json comes out of this script; DESIGN.md
quotes them and says so (round 6: no generated sections inside DESIGN.md any more)."""
import collections
import csv
import glob
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "gpurun_out", "r06e")
F = os.path.join(R, "gpurun_out", "r06f")
P = os.path.join(R, "profiles")
SIMDS = 1024.0
nan = float("nan")


def last_json(path):
    return json.loads([ln for ln in open(path).read().strip().splitlines() if ln.startswith("{")][-1])


# ---- bench lines ----------------------------------------------------------------------------------------------------
lines = {}
for f in sorted(glob.glob(os.path.join(E, "bench_*.json"))):
    try:
        lines[os.path.basename(f)[6:-5]] = last_json(f)
    except Exception as e:
        print("skip", f, e)
final = os.path.exists(os.path.join(F, "bench_driverflags.json"))
if final:
    lines["driverflags_first_call"] = lines["driverflags"]
    lines["driverflags"] = last_json(os.path.join(F, "bench_driverflags.json"))
    for f in sorted(glob.glob(os.path.join(F, "bench_*.json"))):
        k = os.path.basename(f)[6:-5]
        if k != "driverflags":
            lines[k + "_second_call"] = last_json(f)
head = lines["driverflags"]
open(os.path.join(P, "r06_bench_n1.json"), "w").write(json.dumps(head) + "\n")
with open(os.path.join(P, "r06_bench_lines.jsonl"), "w") as fo:
    for k, d in lines.items():
        d = dict(d)
        d["_run"] = k
        fo.write(json.dumps(d) + "\n")
lib_sha = head["roofline"]["library_sha16"]  # (kernel sources + ROCm release: bench.py library_sha16(); both calls ran the same tree)
what = collections.OrderedDict([
    ("driverflags", "cfg2 (64 ch, D = 96, 128 taps), driver's flags"), ("default", "cfg2, defaults (300 steps)"),
    ("grid64", "cfg2 geometry, every channel on the 12.5 kHz raster"), ("mfma1", "cfg2, first-generation kernel"),
    ("c128", "128 channels (configs[2] shard of 8)"), ("c256", "256 channels"), ("c1024", "1024 channels on one GPU (north star's shape)"),
    ("c128_slice128", "128 channels on 128-channel slices (MFM_F_SLICE_128: long-filter kernel, two row blocks per wave)"),
    ("c256_slice128", "256 channels on 128-channel slices"), ("c1024_slice128", "1024 channels on 128-channel slices"),
    ("c128_slice64", "128 channels on 64-channel slices (MFM_F_SLICE_64: mfm_kernel_v3.hip)"), ("c256_slice64", "256 channels on 64-channel slices"),
    ("c1024_slice64", "1024 channels on 64-channel slices (round 5's form of north star's shape)"),
    ("cfg5_auto", "configs[4] per-GPU share: 256 ch, D = 400, 512 taps - long-filter kernel, two row blocks per wave"),
    ("cfg5_v3l1", "the same, one row block per wave forced"), ("cfg5_mfma1", "the same, first generation (rounds 3-4)"),
    ("d25_auto", "pocsag_rtlsdr + its 256-tap file: 64 ch, D = 25 - long-filter kernel"), ("d25_mfma1", "the same, first generation"),
    ("d100_auto", "pocsag_airspy: 64 ch, D = 100, 256 taps - long-filter kernel"), ("d100_mfma1", "the same, first generation"),
    ("d120_auto", "multifm_airspy: 64 ch, D = 120, 512 taps - long-filter kernel"), ("d120_mfma1", "the same, first generation"),
    ("t512_auto", "64 ch, D = 96, 512 taps - long-filter kernel"), ("t512_mfma1", "the same, first generation"),
    ("t256_auto", "64 ch, D = 96, 256 taps - long-filter kernel"), ("t256_mfma1", "the same, first generation"),
    ("pocsag_d25", "etc/pocsag_rtlsdr.json geometry: 64 ch, D = 25, 128 taps"), ("multifm_d40", "etc/multifm.json geometry: 64 ch, 1 MS/s, D = 40")])
hdr = "run                kernel                      value(MSamp/s x ch)  ms/step  kernel_ms  min     median  p95     hbm_frac  sclk(MHz)  verified"
rows = [hdr]
table_md = ["| run | kernel | value (MSamp/s x ch) | ms per step | kernel ms (min / median / p95) | roofline.frac | shader clock in the launches | verified |",
            "|---|---|---|---|---|---|---|---|"]
for k in list(what) + [k for k in lines if k not in what]:
    if k not in lines:
        continue
    d = lines[k]
    r = d["roofline"]
    clk = (r.get("clocks") or {}).get("sclk_mhz_effective", nan)
    rows.append(f"{k:18s} {r['kernel']:27s} {d['value']:14.4g}      {d['ms_per_step']:.4f}   {r['kernel_ms']:.4f}   {r['kernel_ms_min']:.4f}  "
                f"{r['kernel_ms_median']:.4f}  {(r['kernel_ms_p95'] or nan):.4f}  {r['frac']:.3f}     {clk:7.0f}    {d.get('verified')}")
    table_md.append(f"| {what.get(k, k)} | {r['kernel'].replace('mfm_channel_kernel', 'kernel')} | {d['value'] / 1e6:.1f} M | {d['ms_per_step']:.4f} | "
                    f"{r['kernel_ms']:.4f} ({r['kernel_ms_min']:.4f} / {r['kernel_ms_median']:.4f} / {(r['kernel_ms_p95'] or nan):.4f}) | {r['frac']:.3f} | "
                    f"{clk:.0f} MHz | {d.get('verified')} |")
open(os.path.join(P, "r06_bench_table.txt"), "w").write("\n".join(rows) + "\n")
print("\n".join(rows))

# ---- rocprofv3 kernel stats of the headline command -------------------------------------------------------------------
trace = {}
for tag in ("kstats", "kstats1024"):
    for f in glob.glob(os.path.join(E, tag, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, f"r06_rocprofv3_kernel_stats{'' if tag == 'kstats' else '_1024ch'}.csv"))
    tr = glob.glob(os.path.join(E, tag, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rr = sorted([r for r in csv.DictReader(open(tr[0])) if "channel_kernel" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
        du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rr]
        gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000 for a, b in zip(rr[:-1], rr[1:])]
        trace[tag] = {"launches": len(du), "last20_us": sum(du[-20:]) / 20, "all_us": sum(du) / len(du), "name": rr[0]["Kernel_Name"],
                      "last20_gap_us": sum(gaps[-19:]) / 19}
        try:
            trace[tag]["bench"] = last_json(os.path.join(E, tag + ".log"))
        except Exception:
            pass
        print(tag, {k: v for k, v in trace[tag].items() if k != "bench"})
        if tag == "kstats":
            open(os.path.join(P, "r06_kernel_duration_series.txt"), "w").write(
                "# launch durations (us) of the channel kernel in the rocprofv3 trace of `bench.py --gpus 1 --steps 20 --warmup 5`:\n"
                "# settle phase first, the last 25 launches are warm-up + timed region\n" + "\n".join("%.1f" % x for x in du) + "\n")


def stats_avg(path, needle):
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1000, int(r["Calls"]), r["Name"]
    return None


ks = stats_avg(os.path.join(P, "r06_rocprofv3_kernel_stats.csv"), "channel_kernel")
ks1024 = stats_avg(os.path.join(P, "r06_rocprofv3_kernel_stats_1024ch.csv"), "channel_kernel")

# ---- PMC passes, per shape ----------------------------------------------------------------------------------------------
shapes = collections.OrderedDict([("head", "driverflags"), ("c1024", "c1024"), ("c1024s128", "c1024_slice128"), ("c1024s64", "c1024_slice64"),
                                  ("c1024wb", "c1024"),
                                  ("cfg5", "cfg5_auto"), ("d25", "d25_auto"),
                                  ("d100", "d100_auto"), ("d120", "d120_auto")])
pmc, raw, traffic, issue_shapes, pmc_lines = {}, [], {}, {}, {}
for tag, line_key in shapes.items():
    acc = collections.defaultdict(list)
    kn = ""
    for p in ("p1", "p2", "p3", "fetch", "write"):
        for f in glob.glob(os.path.join(E, f"{tag}_{p}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "channel_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if tag == "head" and p in ("fetch", "write"):
                shutil.copy(f, os.path.join(P, f"r06_pmc_{'FETCH' if p == 'fetch' else 'WRITE'}_SIZE.csv"))
    if not acc:
        continue
    m = {}
    raw.append(f"## {tag}: {kn[:110]}")
    for k, v in sorted(acc.items()):
        v = v[len(v) // 2:]
        m[k] = sum(v) / len(v)
        raw.append(f"{k:28s} launches={len(v):3d} mean={m[k]:.6g}")
    pmc[tag] = (m, kn)
    try:
        pmc_lines[tag] = last_json(os.path.join(E, f"{tag}_p3.log"))
    except Exception:
        pass
    d = lines.get(line_key)
    if d and "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        alg = d["roofline"]["bytes_per_launch"]
        t = {"kernel": kn, "workload": d["config"]["workload"], "FETCH_SIZE_kb_per_launch": m["FETCH_SIZE"],
             "WRITE_SIZE_kb_per_launch": m["WRITE_SIZE"],
             "correction": "gfx950: FETCH_SIZE counts 16 B/lane streaming reads at half their bytes (MI355X_MICROARCH.md): x2",
             "hbm_bytes_per_launch": (m["FETCH_SIZE"] * 2 + m["WRITE_SIZE"]) * 1024, "algorithmic_bytes_per_launch": alg}
        t["ratio"] = t["hbm_bytes_per_launch"] / alg
        traffic[tag] = t
    if d and "SQ_INSTS_MFMA" in m and "SQ_INSTS_VALU" in m:
        inst = d["roofline"].get("instance")
        if inst and tag != "c1024wb":  # (slices of 64 and of 128 have keys of their own; the write-back run shares the default's)  # (the forced forms share the default's instance key)
            issue_shapes[inst] = {"tag": tag, "kernel": kn, "workload": d["config"]["workload"], "mfma_insts_per_launch": m["SQ_INSTS_MFMA"],
                                  "other_valu_insts_per_launch": m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"],
                                  "launch_cycles_in_profiled_run": m.get("GRBM_GUI_ACTIVE", nan) / 8.0,
                                  "mfma_busy_cycles_per_simd": m.get("SQ_VALU_MFMA_BUSY_CYCLES", nan) / SIMDS}
open(os.path.join(P, "r06_rocprofv3_pmc_raw.txt"), "w").write("\n".join(raw) + "\n")
if "head" in traffic:
    json.dump(traffic["head"], open(os.path.join(P, "r06_hbm_traffic.json"), "w"), indent=1)
if "c1024" in traffic:
    json.dump(traffic["c1024"], open(os.path.join(P, "r06_hbm_traffic_1024ch.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(P, "r06_hbm_traffic_shapes.json"), "w"), indent=1)
ns_inst = head.get("north_star_shape", {}).get("instance")
json.dump({"source": "profiles/r06_rocprofv3_pmc_raw.txt (tools/prof_r06.sh, tools/collect_r06.py): SQ_INSTS_MFMA and SQ_INSTS_VALU - SQ_INSTS_MFMA per "
                     "launch; keys are bench.py's instance_name() of the (kernel, geometry)",
           "library_sha16": lib_sha, "shapes": issue_shapes}, open(os.path.join(P, "r06_issue_model.json"), "w"), indent=1)
print("issue model instances:", list(issue_shapes), "north star instance in the line:", ns_inst)


def outputs_of(d):
    c = d["config"]
    return c["channels_per_gpu"] * (c["block_samples"] // c["decimation"])


summ = ["# rocprofv3 --pmc summary, round 6 (tools/prof_r06.sh: `bench.py --steps 8 --warmup 3 --settle-seconds 0.3 <shape>`, block 2^26);",
        "# mean per launch over the second half of the profiled launches.  GENERATED by tools/collect_r06.py from r06_rocprofv3_pmc_raw.txt -",
        f"# every figure below is computed from the counters in that file, none is typed.  Kernel sources sha256[:16] = {lib_sha}.", "#"]
pmc_rows = {}
for tag, (m, kn) in pmc.items():
    d = lines.get(shapes[tag])
    if not d or "GRBM_GUI_ACTIVE" not in m:
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    other = m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]
    wc = m.get("SQ_WAVE_CYCLES", nan)
    lane = 64.0 * other / outputs_of(d)
    mf, v3 = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 3.0 * other / (SIMDS * cyc)
    pmc_rows[tag] = {"lane": lane, "mfma": mf, "valu3": v3, "busy": mf + v3, "cycles": cyc, "wait_any": m.get("SQ_WAIT_ANY", nan) / wc,
                     "wait_inst": m.get("SQ_WAIT_INST_ANY", nan) / wc, "kernel": kn,
                     "traffic_ratio": traffic.get(tag, {}).get("ratio", nan),
                     "kernel_ms_profiled": (pmc_lines.get(tag) or {}).get("roofline", {}).get("kernel_ms", nan),
                     "kernel_ms": d["roofline"]["kernel_ms"]}
    summ += [f"## {tag}: {d['config']['workload']}",
             f"#   kernel                               {kn}",
             f"#   launch length                        GRBM_GUI_ACTIVE / 8 XCDs = {cyc:.4g} cycles (serialized by the profiler: "
             f"{pmc_rows[tag]['kernel_ms_profiled']:.4f} ms by the engine's events in that run; {d['roofline']['kernel_ms']:.4f} ms back to back in the un-profiled run)",
             f"#   SQ_INSTS_VALU (incl. MFMA)           {m['SQ_INSTS_VALU']:.4g}   SQ_INSTS_MFMA {m['SQ_INSTS_MFMA']:.4g}   other VALU {other:.4g} "
             f"= {lane:.1f} lane-instructions per (channel, output)",
             f"#   matrix pipe busy                     SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch) = {100 * mf:.1f} %",
             f"#   other VALU                           {100 * v3 * 2 / 3:.1f} / {100 * v3:.1f} / {100 * v3 * 4 / 3:.1f} % at 2 / 3 / 4 cycles per instruction",
             f"#   busy (matrix + other VALU at 3)      {100 * (mf + v3):.1f} %",
             f"#   MFMA time with a VALU instruction beside it   {100 * m.get('SQ_VALU_MFMA_COEXEC_CYCLES', nan) / m['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} %",
             f"#   waves waiting (any reason)           SQ_WAIT_ANY / SQ_WAVE_CYCLES = {100 * m.get('SQ_WAIT_ANY', nan) / wc:.0f} %; for an issue slot "
             f"{100 * m.get('SQ_WAIT_INST_ANY', nan) / wc:.0f} %; for LDS {100 * m.get('SQ_WAIT_INST_LDS', nan) / wc:.1f} %",
             f"#   LDS bank conflicts                   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = "
             f"{100 * m.get('SQ_LDS_BANK_CONFLICT', nan) / m.get('SQ_LDS_IDX_ACTIVE', nan):.0f} %",
             f"#   LDS instructions                     {64.0 * m.get('SQ_INSTS_LDS', nan) / outputs_of(d):.2f} lane-instructions per (channel, output)",
             f"#   instructions                         SALU {m.get('SQ_INSTS_SALU', nan):.4g}, LDS {m.get('SQ_INSTS_LDS', nan):.4g}, VMEM loads "
             f"{m.get('SQ_INSTS_VMEM_RD', nan):.4g}, stores {m.get('SQ_INSTS_VMEM_WR', nan):.4g}",
             f"#   HBM traffic (separate passes)        " + (f"{traffic[tag]['hbm_bytes_per_launch'] / 1e6:.1f} MB = {traffic[tag]['ratio']:.3f} x algorithmic "
                                                             f"({traffic[tag]['algorithmic_bytes_per_launch'] / 1e6:.1f} MB)" if tag in traffic else "-"), "#"]
open(os.path.join(P, "r06_rocprofv3_pmc_summary.txt"), "w").write("\n".join(summ + raw) + "\n")
print("\n".join(summ))

# ---- long filters: second generation against first, targets of VERDICT r05 item 2 ----------------------------------------
targets = {"cfg5": 0.205, "d25": 0.63, "d120": 0.18}  # VERDICT r05, "Next round" item 2
lf = ["# Long filters (129-512 taps): the second-generation long-filter kernel (mfm_kernel_v3l.hip, `auto`) against the first-generation",
      "# resident-tap kernel that rounds 3-4 ran them on (`--kernel mfma1`), same box, same call (tools/prof_r06.sh), 2^26-sample blocks,",
      "# 40 steps, kernel ms by the engine's HIP events.  Counters from the --pmc passes of the same call (r06_rocprofv3_pmc_summary.txt).",
      "# GENERATED by tools/collect_r06.py; the A/B history of how the kernel got here follows below (tools/r06/long_history.txt).", "#",
      "shape  geometry                                   v3l ms   v1 ms    ratio   target  met   lane-instr/(ch,out)  matrix busy  busy(3-cyc)  HBM ratio"]
geo = {"cfg5": "256 ch, D = 400, 512 taps (configs[4])", "d25": "64 ch, D = 25, 256 taps", "d100": "64 ch, D = 100, 256 taps",
       "d120": "64 ch, D = 120, 512 taps", "t512": "64 ch, D = 96, 512 taps", "t256": "64 ch, D = 96, 256 taps"}
lf_md = ["| shape | long-filter kernel (ms) | first generation (ms) | ratio | VERDICT r05 target | lane-instr per (ch, out) | matrix busy | busy (3-cycle) |", "|---|---|---|---|---|---|---|---|"]
for s in ("cfg5", "d25", "d100", "d120", "t512", "t256"):
    a, b = lines.get(s + "_auto"), lines.get(s + "_mfma1")
    if not a or not b:
        continue
    ka, kb = a["roofline"]["kernel_ms"], b["roofline"]["kernel_ms"]
    pr = pmc_rows.get(s, {})
    tg = targets.get(s)
    lf.append(f"{s:6s} {geo[s]:42s} {ka:.4f}   {kb:.4f}   {ka / kb:.3f}   {('%.2f' % tg) if tg else '  - '}    {('yes' if ka <= tg else 'NO ') if tg else ' - '}   "
              f"{pr.get('lane', nan):8.1f}             {100 * pr.get('mfma', nan):5.1f} %      {100 * pr.get('busy', nan):5.1f} %      {pr.get('traffic_ratio', nan):.3f}")
    lf_md.append(f"| {geo[s]} | **{ka:.4f}** | {kb:.4f} | {ka / kb:.3f} | {('≤ %.2f: %s' % (tg, 'met' if ka <= tg else 'not met')) if tg else '-'} | "
                 f"{pr.get('lane', nan):.1f} | {100 * pr.get('mfma', nan):.1f} % | {100 * pr.get('busy', nan):.1f} % |")
if "cfg5_v3l1" in lines:
    lf.append(f"cfg5 with one row block per wave forced (MFM_F_V3L_ONE_ROW_BLOCK): {lines['cfg5_v3l1']['roofline']['kernel_ms']:.4f} ms")
hist = os.path.join(R, "tools", "r06", "long_history.txt")
open(os.path.join(P, "r06_long_filters.txt"), "w").write("\n".join(lf) + "\n\n" + (open(hist).read() if os.path.exists(hist) else ""))
print("\n".join(lf))

for src, dst in (("step_overheads.txt", "r06_step_overheads.txt"), ("link_probe.txt", "r06_link_probe.txt"), ("driver_run_s.txt", "r06_driver_run_s.txt")):
    if os.path.exists(os.path.join(E, src)):
        shutil.copy(os.path.join(E, src), os.path.join(P, dst))

# ---- generated text ---------------------------------------------------------------------------------------------------
hr = head["roofline"]
cb = head.get("cpu_baseline", {})
ns = head.get("north_star_shape", {})
gp = head.get("group_path", {})
gen = [f"Generated by `tools/collect_r06.py` from `gpurun_out/r06e` (`tools/prof_r06.sh`, one box, one `gpurun` call)"
       + (" and `gpurun_out/r06f` (`tools/prof_r06_final.sh`: the driver's command again on the same library, now with `r06_issue_model.json` in place)" if final else "")
       + f"; kernel sources sha256[:16] `{lib_sha}`.  Edit the script, not this text.", ""]
gen.append(f"* Headline (`r06_bench_n1.json`, the driver's command `python bench.py --gpus 1 --steps 20 --warmup 5`): "
           f"**{head['value'] / 1e6:.1f} M MSamp/s x channels**, `ms_per_step` {head['ms_per_step']:.4f}, kernel {hr['kernel_ms']:.4f} ms "
           f"(HIP events on {hr.get('timed_launches')} of {hr.get('launches')} timed launches; all {((hr.get('clocks') or {}).get('kernel_ms_by_stamps') or {}).get('launches', '?')} by the kernel's "
           f"own stamps: mean {((hr.get('clocks') or {}).get('kernel_ms_by_stamps') or {}).get('mean', nan):.4f} ms, min {((hr.get('clocks') or {}).get('kernel_ms_by_stamps') or {}).get('min', nan):.4f}, "
           f"max {((hr.get('clocks') or {}).get('kernel_ms_by_stamps') or {}).get('max', nan):.4f}, first workgroup's start to last workgroup's end), `roofline.frac` **{hr['frac']:.3f}**, "
           f"`verified` {head.get('verified')}.")
clk = hr.get("clocks") or {}
bs = hr.get("board_sample") or {}
bs_note = "" if bs.get("matched") else (" (that sample predates the matching of the card to the device's PCI address and the wait for a sustained load: "
                                          "it is the first AMD card of that box, read right after the host's pause - a matched, sustained reading is in "
                                          "`r06b_summary.txt`: 1944 MHz, 1384 W of the 1400 W cap)")
gen.append(f"* Clock and power of the same run: the kernel's own stamps (`roofline.clocks`: s_memtime / s_memrealtime of the first and last workgroup) give "
           f"{clk.get('shader_ticks_median', nan):.0f} shader cycles in {clk.get('kernel_ms_by_ref_ticks', nan):.4f} ms = **{clk.get('sclk_mhz_effective', nan):.0f} MHz** "
           f"inside the timed launches; the board's sysfs reading ({'the card at the device PCI address, ' + str(bs.get('when')) if bs.get('matched') else 'right behind the timed region'}): {bs.get('sclk_mhz')} MHz, {bs.get('power_w')} W" + (f" of the {bs.get('power_cap_w'):.0f} W cap" if bs.get('power_cap_w') else "") + f"{bs_note}.")
im = hr.get("issue_model")
if im and im.get("simd_busy_fraction"):
    gen.append(f"* Issue model in the line (`roofline.issue_model`): {im['mfma_insts_per_launch']:.4g} matrix + {im['other_valu_insts_per_launch']:.4g} other vector "
               f"instructions per launch (this library, this instance) = {im['issue_cycles_per_simd']:.0f} issue cycles per SIMD against {im['launch_shader_cycles']:.0f} "
               f"shader cycles of THIS run: SIMDs busy **{im['simd_busy_fraction']:.3f}**, `ceiling_frac` {im['ceiling_frac']:.3f}.")
if ks:
    gen.append(f"* `rocprofv3 --kernel-trace --stats` of the same command (`r06_rocprofv3_kernel_stats.csv`): `{ks[2][:60]}` averages "
               f"**{ks[0]:.1f} us** over {ks[1]} launches (settle phase included); the last 20 launches of the trace (the timed region) average "
               f"{trace.get('kstats', {}).get('last20_us', nan):.1f} us with {trace.get('kstats', {}).get('last20_gap_us', nan):.1f} us between one launch's end and the next one's start "
               f"(`r06_kernel_duration_series.txt`); the engine's events in that profiled run: "
               f"{trace.get('kstats', {}).get('bench', {}).get('roofline', {}).get('kernel_ms', nan) * 1e3:.1f} us.  `roofline.frac` uses the un-profiled run's "
               f"back-to-back launches ({hr['kernel_ms'] * 1e3:.1f} us).")
if "head" in pmc_rows:
    pr = pmc_rows["head"]
    gen.append(f"* SQ counters, headline (`r06_rocprofv3_pmc_summary.txt`): {pr['lane']:.1f} lane-instructions per (channel, output), matrix pipe busy "
               f"{100 * pr['mfma']:.1f} %, other VALU {100 * pr['valu3']:.1f} % (3 cycles each), together {100 * pr['busy']:.1f} %; the counter passes serialize the launches "
               f"({pr['kernel_ms_profiled']:.4f} ms per launch there against {pr['kernel_ms']:.4f} back to back).")
if "head" in traffic:
    t = traffic["head"]
    gen.append(f"* HBM traffic (`r06_hbm_traffic.json`; FETCH_SIZE and WRITE_SIZE each in its own `--pmc` pass, FETCH_SIZE doubled for gfx950): "
               f"**{t['hbm_bytes_per_launch'] / 1e6:.1f} MB = {t['ratio']:.3f} x algorithmic** ({t['algorithmic_bytes_per_launch'] / 1e6:.1f} MB).")
if ns and "roofline" in ns:
    nim = ns.get("issue_model") or {}
    gen.append(f"* North star's shape in the line (`north_star_shape`: 1024 channels on the one GPU): kernel **{ns['kernel_ms']:.3f} ms**, "
               f"{ns['value'] / 1e6:.1f} M MSamp/s x channels, `roofline.frac` {ns['roofline']['frac']:.3f} against a matrix-instruction bound of "
               f"{ns['bound_frac']['at_nominal_5000_tops']:.3f} (nominal int8 peak) / {ns['bound_frac']['at_measured_3944_tops']:.3f} (the guide's measured peak); "
               f"{(ns.get('clocks') or {}).get('sclk_mhz_effective', nan):.0f} MHz inside the launches"
               + (f"; SIMDs busy {nim['simd_busy_fraction']:.3f} by the issue model" if nim.get("simd_busy_fraction") else "") + "."
               + (f"  HBM traffic {traffic['c1024']['ratio']:.3f} x algorithmic (`r06_hbm_traffic_1024ch.json`)." if "c1024" in traffic else "")
               + (f"  rocprofv3 trace: {ks1024[0] / 1000:.3f} ms per launch (`r06_rocprofv3_kernel_stats_1024ch.csv`)." if ks1024 else ""))
if gp and "ratio_to_value" in gp:
    gen.append(f"* The group path in the same line (`group_path`: the same blocks through `mfm_group_acquire_input` / `mfm_group_submit`): "
               f"{gp['value'] / 1e6:.1f} M, {gp['ratio_to_value']:.3f} x `value`, verified {gp.get('verified')}.")
if cb:
    gen.append(f"* CPU baseline in the same run: {cb.get('value', 0):.0f} MSamp/s x channels on {cb.get('cores')} threads of a "
               f"{cb.get('host_cores')}-core host ({cb.get('host_cpu')}); one channel on one core: {cb.get('msamp_per_s_one_channel_one_core', 0):.0f} MSamp/s.")
spread = [lines[k]["roofline"]["kernel_ms"] for k in ("driverflags", "driverflags_first_call", "default", "default_second_call") if k in lines]
if len(spread) > 1:
    gen.append(f"* The same kernel, same box, {len(spread)} runs of this collection: {', '.join('%.4f' % x for x in spread)} ms "
               f"(spread {100 * (max(spread) / min(spread) - 1):.1f} %).")
gen += ["", "Long filters (`r06_long_filters.txt`):", ""] + lf_md + ["", "All runs (`r06_bench_table.txt`, `r06_bench_lines.jsonl`):", ""] + table_md
gen_text = "\n".join(gen)
open(os.path.join(P, "r06_summary.md"), "w").write(gen_text + "\n")

vals = {"ms_step": f"{head['ms_per_step']:.4f}", "kernel_ms": f"{hr['kernel_ms']:.4f}", "frac": f"{hr['frac']:.3f}",
        "ns_ms": f"{ns.get('kernel_ms', nan):.3f}", "ns_frac": f"{ns.get('roofline', {}).get('frac', nan):.3f}"}
for s in ("cfg5", "d25", "d100", "d120", "t512", "t256"):
    if s + "_auto" in lines and s + "_mfma1" in lines:
        vals[s + "_v3l"] = f"{lines[s + '_auto']['roofline']['kernel_ms']:.4f}"
        vals[s + "_v1"] = f"{lines[s + '_mfma1']['roofline']['kernel_ms']:.4f}"
    if s in pmc_rows:
        vals[s + "_lane"] = f"{pmc_rows[s]['lane']:.1f}"
        vals[s + "_busy"] = f"{pmc_rows[s]['busy']:.2f}"
# ---- block-size series, host-fed figures, the link, latency (from the driver-flags line) --------------------------------------
bs = head.get("block_series", {}).get("series", [])
st = ["| block | mode | launches | us per block | input GSamp/s | of the HBM roof |", "|---|---|---|---|---|---|"]
names = {"coalesced": "backlog gathered into launches of up to 2^26 samples, two streams", "per_block": "every block its own launch, two streams",
         "coalesced_one_stream": "gathered, one stream", "per_block_one_stream": "every block its own launch, one stream"}
for row in bs:
    for mode in ("coalesced", "coalesced_one_stream", "per_block", "per_block_one_stream"):
        m = row.get(mode)
        if isinstance(m, dict) and "frac" in m:
            st.append(f"| 2^{row['block_samples'].bit_length() - 1} x {row['blocks']} | {names[mode]} | {m['launches']} | {m['us_per_block']:.2f} | "
                      f"{m['input_msamp_per_s'] / 1e3:.1f} | **{m['frac']:.3f}** |")
ee = head.get("end_to_end", {})
st += ["", f"Host-fed (`end_to_end`: {ee.get('path')}; PCIe both ways inside the figure).  `pool_arena` modes: the buffers are frames of ONE "
       "page-locked pool in address order, as `host/mfm_receiver.c` gets them from its frame pool, and runs of neighbours go to the device as one "
       "strided copy command (`mfm_group_push_pinned_run`):", "",
       "| mode | buffers | copy commands | launches | us per buffer | input GSamp/s | H2D GB/s | D2H GB/s |", "|---|---|---|---|---|---|---|---|"]
for mode, m in ee.items():
    if isinstance(m, dict) and "input_msamp_per_s" in m:
        st.append(f"| {mode} | {m.get('buffers', ee.get('buffers'))} | {m.get('copy_commands', '-')} | {m['launches']} | {m['us_per_buffer']:.2f} | "
                  f"{m['input_msamp_per_s'] / 1e3:.2f} | {m['h2d_GBps']:.1f} | {m['d2h_GBps']:.1f} |")
lk = ee.get("link")
if lk:
    st += ["", "The link by itself (`end_to_end.link`: `mfm_link_probe[_runs]` - page-locked host memory, H2D with the D2H of a third of the bytes "
           "running against it, as the path has it):", "", "| pieces | H2D GB/s | D2H GB/s | H2D alone |", "|---|---|---|---|"]
    for k, lab in (("pieces_512KiB", "512 KiB per copy command (one RTL-SDR sample_buf)"), ("pieces_512KiB_runs_of_16_strided", "runs of 16 x 512 KiB as one strided command"),
                   ("pieces_64MiB", "64 MiB per command")):
        if k in lk:
            st.append(f"| {lab} | {lk[k]['h2d_GBps']:.1f} | {lk[k]['d2h_GBps']:.1f} | {lk[k].get('h2d_alone_GBps', nan):.1f} |")
    st.append("")
    st.append("end_to_end over link: " + ", ".join(f"{k[len('end_to_end_'):]} {v:.2f}" for k, v in lk.items() if k.startswith("end_to_end_")))
lat = ee.get("latency")
if lat:
    st += ["", "Latency at a live feed's rate (`end_to_end.latency`: from `deliver` of a 131 072-sample buffer to its PCM fetched on the host, "
           "buffers arriving at the sample rate's pace):", "", "| launch policy | feed | median ms | max ms | buffer period ms |", "|---|---|---|---|---|"]
    for pol, feeds in lat.items():
        for feed, m in feeds.items():
            st.append(f"| {pol} | {feed} | {m['latency_ms_median']:.3f} | {m['latency_ms_max']:.3f} | {m['buffer_period_ms']:.1f} |")
cba = cb.get("all_cores") if cb else None
if cba:
    st += ["", f"CPU baseline on all cores (`cpu_baseline.all_cores`): {cba['value']:.0f} MSamp/s x channels on {cba['cores']} threads, {cba['channels']} channels."]
series_text = "\n".join(st)
open(os.path.join(P, "r06_block_series.md"), "w").write(series_text + "\n")

# the exchange table of DESIGN.md section 7
blk_mb = head["config"]["block_samples"] * 4 / 1e6
xt = ["| channels per GPU | kernel per block | needed per peer (int16 / 8-bit) | broadcast (≈ 153 GB/s per GPU) | all-gather on 7 links (≈ 940 GB/s at N = 8) |",
      "|---|---|---|---|---|"]
for key, lab in (("driverflags", "64"), ("c128", "128 (configs[2]: 1024 on 8 GPUs)"), ("c256", "256"), ("c1024", "1024")):
    if key not in lines:
        continue
    k = lines[key]["roofline"]["kernel_ms"]
    need = blk_mb / k  # MB per ms = GB/s
    fmt = lambda have, n: "hidden" if n <= have else f"{n / have:.1f} x short"
    xt.append(f"| {lab} | {k:.3f} ms | {need:.0f} / {need / 2:.0f} GB/s | {fmt(153.0, need)} | {fmt(940.0, need)} (8-bit: {fmt(940.0, need / 2)}) |")
xt_text = "\n".join(xt)

# round 6: the generated tables live under profiles/ only (DESIGN.md and profiles/README.md point at them; nothing in those two
# files is rewritten by this script any more)
open(os.path.join(P, "r06_exchange_table.md"), "w").write("Needed exchange rate per peer against what the links give, by channels per GPU (GENERATED by tools/collect_r06.py from the "
                                                           "kernel times of profiles/r06_bench_lines.jsonl; DESIGN.md section 7):\n\n" + xt_text + "\n")
json.dump(vals, open(os.path.join(P, "r06_values.json"), "w"), indent=1)
print("wrote profiles/r06_summary.md, r06_block_series.md, r06_exchange_table.md, r06_values.json")
