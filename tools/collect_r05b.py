#!/usr/bin/env python3
"""gpurun_out/r05b (tools/prof_r05b.sh: the driver's command on the round's final bench.py, its rocprofv3 kernel trace, the gpu
test suite and smoke() from ONE box) -> profiles/r05b_bench_n1.json, r05b_rocprofv3_kernel_stats.csv, r05b_summary.txt."""
import csv
import json
import os
import shutil
import statistics

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out", "r05b")
P = os.path.join(R, "profiles")
shutil.copy(os.path.join(G, "bench_driverflags.json"), os.path.join(P, "r05b_bench_n1.json"))
shutil.copy(os.path.join(G, "kstats", "k_kernel_stats.csv"), os.path.join(P, "r05b_rocprofv3_kernel_stats.csv"))
d = json.loads(open(os.path.join(G, "bench_driverflags.json")).read().strip().splitlines()[-1])
r = d["roofline"]
c = r["clocks"]
b = r["board_sample"]
k = list(csv.DictReader(open(os.path.join(G, "kstats", "k_kernel_stats.csv"))))[0]
tr = [x for x in csv.DictReader(open(os.path.join(G, "kstats", "k_kernel_trace.csv"))) if x["Kernel_Name"].startswith(k["Name"][:40])]
du = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in tr]
passed = [ln for ln in open(os.path.join(G, "gputest.txt")).read().splitlines() if " passed" in ln or " failed" in ln][-1].strip()
smoke = open(os.path.join(G, "smoke.txt")).read().strip().splitlines()[-1]
st = c["kernel_ms_by_stamps"]
ns = d["north_star_shape"]
txt = f"""# Round 5, second session: the driver's command on the round's FINAL bench.py, its rocprofv3 kernel trace, the gpu test suite and
# smoke() - one box, one gpurun call (tools/prof_r05b.sh, tools/collect_r05b.py); kernel sources {r.get('library_sha16')} as in profiles/r05_*.
#
# python bench.py --gpus 1 --steps 20 --warmup 5   (profiles/r05b_bench_n1.json; {open(os.path.join(G, 'driver_run_s.txt')).read().strip()})
value {d['value'] / 1e6:.2f} M MSamp/s x channels   ms_per_step {d['ms_per_step']:.4f}   kernel {r['kernel_ms']:.4f} ms (HIP events, {r['timed_launches']} of {r['launches']} launches)
kernel by its own stamps, all {st['launches']} timed launches: mean {st['mean']:.4f} min {st['min']:.4f} max {st['max']:.4f} ms; {c['shader_ticks_median']:.0f} shader cycles; {c['sclk_mhz_effective']:.0f} MHz inside the launches
roofline.frac {r['frac']:.3f}   ceiling_frac {r.get('ceiling_frac'):.3f}   verified {d['verified']} (the last timed launch)   group_path ratio {d['group_path']['ratio_to_value']:.3f}
board_sample - the card at the device's PCI address {b['pci_address']} (matched {b['matched']}, {b['cards_visible']} cards visible), {b['when']}:
   {b['sclk_mhz']:.0f} MHz, PPT {b['power_w']:.0f} W of the {b['power_cap_w']:.0f} W cap ({100 * b['power_of_cap']:.1f} %)
north_star_shape (1024 channels): kernel {ns['kernel_ms']:.4f} ms, frac {ns['roofline']['frac']:.3f}
#
# rocprofv3 --kernel-trace --stats of the same command (profiles/r05b_rocprofv3_kernel_stats.csv):
{k['Name'][:60]}: {int(k['Calls'])} launches (settle phase and the sustained-load loop behind the timed region included), average {float(k['AverageNs']) / 1e3:.1f} us, min {int(k['MinNs']) / 1e3:.1f}, {float(k['Percentage']):.1f} % of GPU time
algorithmic bytes 357.9 MB / {float(k['AverageNs']) / 1e3:.1f} us / 8 TB/s = {357913939.2 / float(k['AverageNs']) / 8000:.3f}
#
# pytest tests -m gpu -x -q on that box: {passed}
# __graft_entry__.smoke(): {smoke}
"""
open(os.path.join(P, "r05b_summary.txt"), "w").write(txt)
print(txt)
