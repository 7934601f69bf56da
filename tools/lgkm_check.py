#!/usr/bin/env python3
"""Are the hand-counted s_waitcnt lgkmcnt(N) of a kernel's inline-asm LDS reads enough?  tools/lgkm_check.py <object> <filter>

The compiler's waitcnt insertion does not see LDS reads issued from inline asm (the resident long-filter instances of
mfm_kernel_mfma.hip and the hand-scheduled column groups of mfm_kernel_v3.hip request their B fragments that way and wait
with explicit, counted s_waitcnt).  This walks the disassembly of every kernel whose demangled name contains <filter>
and models the LGKM counter: LDS operations complete in order, so after `s_waitcnt lgkmcnt(N)` only the N youngest are
still outstanding.  Any instruction that reads or writes a destination register of a ds_read that is still outstanding is
reported - a copy the register allocator slipped in, a product scheduled in front of its wait.  Straight-line model: the
queue is cleared at labels and branches (the matrix phases are fully unrolled).  Scalar memory loads share the counter
and may return out of order; a counted wait with one of them outstanding is reported as unverifiable.
Prints one line per kernel: reads checked, violations.  Exit code 1 on a violation."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + d + "/fat.bin", obj])
        subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + d + "/fat.bin", "--output=" + d + "/k.co", "--unbundle"])
        return subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", d + "/k.co"], capture_output=True, text=True,
                              check=True).stdout


def regs_of(text):
    """VGPR numbers an operand string mentions (v12, v[12:15]); AGPRs are not used by these kernels"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check_kernel(lines):
    queue = []        # outstanding LGKM operations, oldest first: (kind, dst register set)
    reads = viol = unverifiable = 0
    notes = []
    for ln in lines:
        ins = ln.strip()
        if not ins or ins.endswith(":"):
            queue = []
            continue
        op = ins.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            queue = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                if n and any(k == "smem" for k, _ in queue):
                    unverifiable += 1
                while len(queue) > n:
                    queue.pop(0)
            continue
        # an instruction that touches a register some outstanding ds_read will still write
        pending = set().union(*[d for k, d in queue if k == "ds_read"]) if queue else set()
        if pending:
            hit = regs_of(ins.split(None, 1)[1] if " " in ins else "") & pending
            if hit and not op.startswith("ds_read"):   # a later ds_read into the same slot is ordered behind the earlier one
                viol += 1
                if len(notes) < 5:
                    notes.append(f"{ins}  touches v{sorted(hit)[:4]} before its wait")
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dst = regs_of(ins.split(None, 1)[1].split(",")[0])
            queue.append(("ds_read", dst))
            reads += 1
        elif op.startswith("ds_"):
            queue.append(("ds_other", set()))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append(("smem", set()))
    return reads, viol, unverifiable, notes


def main():
    obj, flt = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    text = disassemble(obj)
    bad = 0
    cur, name = [], None
    kernels = []
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            if name:
                kernels.append((name, cur))
            name, cur = m.group(1), []
        elif name is not None:
            cur.append(ln.split("//")[0])
    if name:
        kernels.append((name, cur))
    for name, lines in kernels:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void ", "", dem).split("(")[0]
        if flt not in dem:
            continue
        reads, viol, unv, notes = check_kernel(lines)
        print(f"{dem:72s} lds reads {reads:4d}  violations {viol}  unverifiable waits {unv}")
        for n in notes:
            print("    " + n)
        bad += viol
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
