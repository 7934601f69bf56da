#!/usr/bin/env python3
"""Turn gpurun_out/r01b + gpurun_out/pmc (tools/prof_r01.sh, tools/pmc_r01.sh) into the files kept under profiles/."""
import collections, csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = os.path.join(R, "gpurun_out", "r01b")
d = json.loads(open(os.path.join(b, "bench.json")).read().strip().splitlines()[-1])
print("bench:", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["compute_roofline"]["frac"],
      d.get("cpu_baseline", {}).get("value"))
rows = list(csv.DictReader(open(os.path.join(b, "kstats", "k_kernel_trace.csv"))))
t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mfma" in r["Kernel_Name"])
du = [(e - s) / 1000 for s, e in t]
print("rocprof: launches", len(du), "mean all %.1f us, mean of the timed ones (150..) %.1f us" % (sum(du) / len(du), sum(du[150:]) / len(du[150:])))
res = {}
for name, f in (("FETCH_SIZE", "fetch/f_counter_collection.csv"), ("WRITE_SIZE", "write/w_counter_collection.csv")):
    rr = list(csv.DictReader(open(os.path.join(b, f))))
    vals = [float(r["Counter_Value"]) for r in rr if "mfma" in r["Kernel_Name"] and r["Counter_Name"] == name]
    res[name] = sum(vals) / max(1, len(vals))
kname = [r["Kernel_Name"] for r in rows if "mfma" in r["Kernel_Name"]][0]
tr = json.load(open(os.path.join(R, "profiles", "r01_hbm_traffic.json")))
tr.update({"kernel": kname, "FETCH_SIZE_kb_per_launch": res["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": res["WRITE_SIZE"],
           "hbm_bytes_per_launch": (res["FETCH_SIZE"] * 2 + res["WRITE_SIZE"]) * 1024})
json.dump(tr, open(os.path.join(R, "profiles", "r01_hbm_traffic.json"), "w"), indent=1)
print("traffic: fetch %.0f KB write %.0f KB -> %.1f MB = %.3f x algorithmic" % (res["FETCH_SIZE"], res["WRITE_SIZE"], tr["hbm_bytes_per_launch"] / 1e6,
      tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"]))
shutil.copy(os.path.join(b, "kstats", "k_kernel_stats.csv"), os.path.join(R, "profiles", "r01_rocprofv3_kernel_stats.csv"))
open(os.path.join(R, "profiles", "r01_bench_n1.json"), "w").write(json.dumps(d) + "\n")
shutil.copy(os.path.join(b, "fetch", "f_counter_collection.csv"), os.path.join(R, "profiles", "r01_pmc_FETCH_SIZE.csv"))
shutil.copy(os.path.join(b, "write", "w_counter_collection.csv"), os.path.join(R, "profiles", "r01_pmc_WRITE_SIZE.csv"))
pm = os.path.join(R, "gpurun_out", "pmc")
if os.path.isdir(pm):
    out = []
    for p in ("p1", "p2", "p3"):
        f = os.path.join(pm, p, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append(f"## pass {p}")
        out += [f"{k:28s} launches={len(v):2d} mean={sum(v) / len(v):.6g}" for k, v in sorted(acc.items())]
    print("\n".join(out))
    open("/tmp/pmc_summary.txt", "w").write("\n".join(out) + "\n")
