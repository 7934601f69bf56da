"""The C-ABI library loads, exports everything include/multifm_hip.h declares, validates arguments the
way the reference's TSL_ASSERT_ARG checks do, and programs channels with the reference's tap
arithmetic.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "multifm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mfm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libmultifm_hip.so does not export {n}"
    assert sorted(pkg.binding.ABI_SYMBOLS) == names


def test_library_exports_nothing_but_the_declared_symbols(pkg):
    """the kernels' launchers, instance selectors and tap helpers are shared between the library's objects only (a linker
    version script generated from the header, tsl-sdr_amd/Makefile): `nm -D` lists exactly the header's names"""
    import subprocess
    so = os.path.join(ROOT, "tsl-sdr_amd", "libmultifm_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == _declared_functions()


FAKE_RCCL = os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")


def _fake_rccl(tmp_path, name="librccl.so"):
    import subprocess
    d = tmp_path / "private"
    d.mkdir(exist_ok=True)
    so = d / name
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", str(so), FAKE_RCCL], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return so


def _rccl_probe(code, env):
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as ge\npkg = ge.load_package()\n" % ROOT + code],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return r.stdout.strip().splitlines()[-1]


def test_rccl_loader_prefers_a_mapped_library_then_the_search_path(pkg, tmp_path):
    """mfm_group.hip load_rccl(): a librccl that the process has mapped already wins (PyTorch brings its own: a second copy in one
    address space would carry its own device state); else the loader's search for the bare name.  Checked with the test double of
    RCCL on a private path - no device involved: the library is only loaded and asked where it lives."""
    so = _fake_rccl(tmp_path)
    base = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "LD_PRELOAD")}
    # 1. on no search path at all, but mapped by the process (ctypes) before the group asks
    got = _rccl_probe("import ctypes; ctypes.CDLL(%r)\nprint(pkg.binding.rccl_library())" % str(so), dict(base, LD_LIBRARY_PATH=""))
    assert os.path.realpath(got) == os.path.realpath(so)
    # 2. first on LD_LIBRARY_PATH
    got = _rccl_probe("print(pkg.binding.rccl_library())", dict(base, LD_LIBRARY_PATH=str(so.parent)))
    assert os.path.realpath(got) == os.path.realpath(so)
    # 3. named by the operator: wins over a mapped copy (here: a second double under another name, mapped first)
    other = _fake_rccl(tmp_path, "librccl.so.1")
    got = _rccl_probe("import ctypes; ctypes.CDLL(%r)\nprint(pkg.binding.rccl_library())" % str(other),
                      dict(base, LD_LIBRARY_PATH="", MFM_RCCL_LIBRARY=str(so)))
    assert os.path.realpath(got) == os.path.realpath(so)
    # ... and a name that cannot be loaded is an error, not a reason to look elsewhere
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as ge\npkg = ge.load_package()\n"
                        "print(pkg.binding.rccl_library())" % ROOT], capture_output=True, text=True, timeout=300,
                       env=dict(base, LD_LIBRARY_PATH=str(so.parent), MFM_RCCL_LIBRARY=str(tmp_path / "nowhere.so")))
    assert r.returncode != 0 and "MFM_RCCL_LIBRARY" in r.stdout + r.stderr, r.stdout[-800:] + r.stderr[-800:]


def test_error_codes_and_messages(pkg):
    lib = pkg.load_library()
    assert lib.mfm_strerror(0) == b"ok"
    assert b"invalid" in lib.mfm_strerror(pkg.binding.MFM_E_INVAL)
    with pytest.raises(pkg.MfmError) as ei:
        pkg.Engine(2400000, 0, 1 << 16)  # decimation 0: receiver.c:165-170 rejects it too
    assert ei.value.code == pkg.binding.MFM_E_INVAL
    with pytest.raises(pkg.MfmError):
        pkg.Engine(0, 96, 1 << 16)


def test_channel_validation(pkg):
    taps = pkg.synth.design_lpf(128)
    e = pkg.Engine(2400000, 96, 1 << 16)
    assert e.add_channel(101000, taps) == 0
    assert e.add_channel(-37500, taps, gain=2.0) == 1
    with pytest.raises(pkg.MfmError) as ei:
        e.add_channel(0, taps[:64])  # all channels share lpfTaps (receiver.c:175-184)
    assert ei.value.code == pkg.binding.MFM_E_INVAL
    e2 = pkg.Engine(2400000, 96, 1 << 16)
    with pytest.raises(pkg.MfmError):
        e2.add_channel(0, taps[:64])  # taps < decimation: the reference segfaults (direct_fir.c:394-398)
    e3 = pkg.Engine(2400000, 96, 1 << 16)
    cim = np.zeros(128, np.int16)
    cim[5] = -32768
    with pytest.raises(pkg.MfmError):
        e3.add_channel_q14(np.zeros(128, np.int16), cim, (16384, 0))
    # data-path calls before commit are state errors, not crashes
    with pytest.raises(pkg.MfmError) as ei:
        e.acquire_input()
    assert ei.value.code == pkg.binding.MFM_E_STATE
    assert e.push(np.zeros((16, 2), np.int16)) == pkg.binding.MFM_E_STATE


@pytest.mark.parametrize("fs,decim", [(2400000, 96), (1000000, 40), (1200000, 25), (10000000, 400)])
def test_taps_and_rotator_increment_equal_oracle(pkg, ora, fs, decim):
    """mfm_engine_add_channel() must quantise taps exactly like demod.c:232-243 / direct_fir.c:72-77."""
    nt = 512 if decim == 400 else 128
    taps = pkg.synth.design_lpf(nt, 12500.0, fs)
    rng = np.random.RandomState(decim)
    offs = [0, 1, -1, 3125, 101000, -320000, -492000, 112500, fs // 2 - 1, -(fs // 2)] + \
        list(rng.randint(-fs // 2, fs // 2, size=150))
    gains = [1.0, 2.5118864315095806, 0.1, 7.3]
    e = pkg.Engine(fs, decim, 1 << 16)
    want = []
    for i, o in enumerate(offs):
        g = gains[i % len(gains)]
        e.add_channel(int(o), taps, gain=g)
        want.append((ora.make_taps(taps, int(o), fs, g), ora.rot_incr(int(o), fs, decim)))
    for c, ((cre, cim), incr) in enumerate(want):
        gre, gim, gincr = e.get_channel(c)
        assert np.array_equal(gre, cre) and np.array_equal(gim, cim), f"taps differ for offset {offs[c]}"
        assert np.array_equal(gincr, incr)


def test_gain_from_db_convention(ora):
    # receiver.c:218-220: 10^(dB/10) applied to amplitude taps
    assert abs(ora.lib().mfmo_gain_from_db(4.0) - 2.5118864315095806) < 1e-15


def test_commit_without_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    e = pkg.Engine(2400000, 96, 1 << 16)
    e.add_channel(0, pkg.synth.design_lpf(128))
    with pytest.raises(pkg.MfmError) as ei:
        e.commit()
    assert ei.value.code == pkg.binding.MFM_E_DEVICE


def test_stages_without_gpu_fail_loudly(pkg):
    """no CPU path anywhere: resampler, POCSAG, FLEX, Mueller-Muller and the float engine refuse to exist without a device"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    b = pkg.binding
    for make in (lambda: b.Resampler(2, np.array([16384], np.int16), 1, 1, 1024),
                 lambda: b.Pocsag(2, 4096),
                 lambda: b.Flex(2, 4096),
                 lambda: b.F32Engine(2400000, 96, 1 << 16)):
        with pytest.raises(pkg.MfmError) as ei:
            make()
        assert ei.value.code == b.MFM_E_DEVICE
    with pytest.raises(pkg.MfmError):
        b.bch3121_decode(np.zeros(4, np.uint32))


def test_asm_scheduled_instances_do_not_spill():
    """The resident long-filter instances of the first-generation matrix kernel request their LDS fragments by
    inline asm and wait for them explicitly: a fragment register the compiler saved to scratch between the request and the wait
    would save what was in it before the data arrived.  So none of them may spill (tools/kernel_regs.py reads the code
    object's notes; the build leaves the object under tsl-sdr_amd/build)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(root, "tsl-sdr_amd", "build", "mfm_kernel_mfma.o")
    if not os.path.exists(obj) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no built object / no llvm tools here")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_regs.py"), obj, "mfm_channel_kernel_mfma<"],
                         capture_output=True, text=True, check=True).stdout
    resident = [ln for ln in out.splitlines()
                if re.search(r"mfma<(16, false, false, \d, 1, \d+, [12], (true|false)|8, false, false, \d, 1, \d+, [12], false)>", ln)]
    assert len(resident) >= 50, len(resident)
    for ln in resident:
        m = re.search(r"vgpr\s+(\d+) agpr\s+\d+ spill\s+(\d+)", ln)
        assert m and int(m.group(1)) <= 256 and int(m.group(2)) == 0, ln


def test_hand_counted_lds_waits_cover_their_reads():
    """the resident long-filter instances (mfm_kernel_mfma.hip) and the hand-scheduled column groups
    (mfm_kernel_v3.hip) issue ds_read_b128 from inline asm and wait with hand-counted s_waitcnt lgkmcnt(N), which the
    compiler's own waitcnt insertion does not model.  tools/lgkm_check.py models the LGKM counter over the disassembly of
    every instance and reports any instruction that touches a fragment register before the wait that covers its read - a
    copy slipped in by the register allocator, a product scheduled in front of its wait.  None may exist, in any instance."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import lgkm_check
    finally:
        sys.path.pop(0)
    # the checker itself: a copy of a fragment register between the read and its wait is a violation, one behind it is not
    reads, viol, unv, _ = lgkm_check.check_kernel(["ds_read_b128 v[0:3], v10 offset:16", "ds_read_b128 v[4:7], v10 offset:32",
                                                   "v_mov_b32_e32 v20, v5", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v21, v1",
                                                   "v_mov_b32_e32 v22, v6", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v23, v7"])
    assert (reads, viol, unv) == (2, 2, 0)
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm tools here")
    # (round 6: every instruction of a matrix phase of the long-filter kernel is inline asm in an order computed at compile time,
    # mfm_v3l_plan.h - fragment reads, transposition and staging stores alike count on LGKM and are replayed here)
    long_objs = [(f"mfm_kernel_v3l_kq{k}.o", "mfm_channel_kernel_v3l<", 8 if k == 4 else 30) for k in (4, 6, 8, 9, 10, 11, 12, 14, 16)]
    for obj, flt, least in [("mfm_kernel_mfma.o", "mfm_channel_kernel_mfma<", 100), ("mfm_kernel_v3.o", "mfm_channel_kernel_v3<", 60)] + long_objs:
        path = os.path.join(root, "tsl-sdr_amd", "build", obj)
        if not os.path.exists(path):
            pytest.skip("no built object")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "lgkm_check.py"), path, flt], capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("mfm_channel_kernel")]
        assert r.returncode == 0 and len(lines) >= least, (r.returncode, len(lines), r.stdout[-2000:])
        assert all(" violations 0 " in ln for ln in lines)


def test_long_filter_instances_do_not_spill_where_it_would_matter():
    """mfm_kernel_v3l.hip: the instances with one row block per wave must not use scratch at all; those with two (128 tap
    registers) may park chunk set-up values there - outside the matrix phase, which test_hand_counted_lds_waits_cover_their_reads
    checks instruction by instruction (a fragment register spilled between its request and its wait would be a violation)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no llvm tools here")
    seen = 0
    for k in (4, 6, 8, 9, 10, 11, 12, 14, 16):
        path = os.path.join(root, "tsl-sdr_amd", "build", f"mfm_kernel_v3l_kq{k}.o")
        if not os.path.exists(path):
            pytest.skip("no built object")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_regs.py"), path, "mfm_channel_kernel_v3l<"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for ln in r.stdout.splitlines():
            m = re.match(r"mfm_channel_kernel_v3l<(\d+), (\d+), (\d+), (\d+), (true|false), (\d+), (true|false), (true|false)>.*vgpr +(\d+).*scratch +(\d+)", ln)
            assert m, ln
            seen += 1
            # round 6: no instance uses scratch at all (every matrix phase is asm; the accumulators, fragments and staging
            # registers of a phase are live across hundreds of statements the compiler cannot reorder)
            assert int(m.group(10)) == 0, ln
            assert int(m.group(9)) <= 256
    assert seen >= 300
