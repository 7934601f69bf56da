#!/usr/bin/env python3
"""Register and spill counts of the kernel instances in an object file: tools/kernel_regs.py build/x.o [name filter]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + d + "/fat.bin", obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + d + "/fat.bin", "--output=" + d + "/k.co", "--unbundle"])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/k.co"], capture_output=True, text=True).stdout
for ent in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", ent).group(1))
    name = re.search(r"\.name:\s+(\S+)", ent).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void ", "", dem).split("(")[0]
    if flt in dem:
        print("%-70s vgpr %3d agpr %3s spill %3d | sgpr %3d spill %3d | lds %6d scratch %5d" % (
            dem, get("vgpr_count"), ent.split()[0], get("vgpr_spill_count"), get("sgpr_count"), get("sgpr_spill_count"),
            get("group_segment_fixed_size"), get("private_segment_fixed_size")))
