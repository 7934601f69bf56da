#!/usr/bin/env python3
"""bench.py - throughput of the multifm channel hot path on N MI355X GPUs.

One "step" = one pass of the fused FIR + derotator + FM-discriminator kernel over one block of
synthetic wideband int16 IQ (default 2^26 samples = 256 MiB) for every channel a GPU owns.

  N = 1 : BASELINE.json configs[1] - 64 channels, 128-tap 25 kHz LPF, decimation 96, 2.4 MS/s-shaped IQ.
  N > 1 : the same 64 channels PER GPU (weak scaling; N = 8 with --channels-per-gpu 128 is configs[2]).
          The wideband block lives on rank 0 and is broadcast to the other ranks with RCCL
          (torch.distributed backend "nccl") into the engines' input buffers, double-buffered against
          the kernel; each rank demodulates only its own contiguous channel range - no other collective.

Metric (BASELINE.json): input IQ MSamp/s x channels demodulated, whole job.  The JSON line also carries
the HBM roofline of the dominant kernel (algorithmic bytes / HIP-event kernel time), the compute roofline of
the kernel variant in use next to it, and - on rank 0 at N = 1 - the oracle's CPU path timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_DOT2_PEAK = 256 * 4 * 32 * 2.4e9  # lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6e12 (v_dot2 measured at half of it)
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8/fp8 MFMA, MI355X_MICROARCH.md (measured 4.1-4.4 POPS)
MFMA_I8_MEASURED_TOPS = 3944.0  # the guide's measured ceiling for v_mfma_i32_16x16x64_i8 (">= 3944 TOPS")
SIMDS = 1024                # 256 CUs x 4
MFMA_ISSUE_CYCLES = 16      # v_mfma_i32_16x16x64_i8: 4 passes x 4 cycles
VALU_ISSUE_CYCLES = 3       # mean issue cost of the kernels' other vector instructions (profiles/r02_ubench_ops.txt: 2 .. 4)
FP32_VECTOR_PEAK_TFLOPS = 157.3  # packed FP32 on the vector ALUs: 256 CU x 256 flop/clk x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Power management on MI355X: the first ~10 launches of a run take 152 us, the next ~100 climb to 200+ us and
    # decay back, after ~150 launches the kernel sits at its sustained ~150 us (profiles/README.md has the series).
    # The defaults put the timed region behind that transient; the whole default run is still well under a second
    # of GPU time.
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=150)
    # The settle phase makes the line independent of --warmup/--steps: back-to-back launches for at least this much
    # wall time (untimed, before the warm-up), so that a short run (--warmup 5 --steps 20) times the same sustained
    # state as the defaults.
    ap.add_argument("--settle-seconds", type=float, default=0.75)
    ap.add_argument("--channels-per-gpu", type=int, default=None,
                    help="default 64 at every N: BASELINE configs[1] per GPU, the same per-GPU work as N grows (weak scaling); "
                             "--gpus 8 --channels-per-gpu 128 is configs[2] (1024 channels on 8 GPUs)")
    ap.add_argument("--kernel", choices=["auto", "mfma1", "mfma1s", "dot2", "v3l1", "slice64", "slice128"], default="auto",
                    help="mfma1 / dot2 = the first-generation matrix kernel / the v_dot2 kernel through the "
                         "MFM_F_FORCE_* flags (A/B timing)")
    ap.add_argument("--pcm-write-back", action="store_true", help="MFM_F_PCM_WRITE_BACK: no system-scope PCM stores (A/B timing, traffic)")
    ap.add_argument("--block-log2", type=int, default=26, help="log2 of wideband samples per step")
    ap.add_argument("--config", default="cfg2_64ch", help="plan name in tsl-sdr_amd/synth.py")
    ap.add_argument("--overlap", action="store_true", help="MFM_F_OVERLAP: consecutive launches on two compute streams")
    ap.add_argument("--input", choices=["cs16", "rtlsdr_u8"], default="cs16",
                    help="rtlsdr_u8: the wideband block is 8-bit IQ as an RTL-SDR delivers it (multifm/rtl_sdr_if.c:146-148); the matrix "
                         "kernel reads the bytes and, at N > 1, the exchange moves half the bytes")
    ap.add_argument("--exchange", choices=["group", "torch"], default="group",
                    help="N > 1: 'group' = ONE process drives the N devices through the library's own device group (mfm_group_*: channel "
                         "shards, the C library's RCCL scatter + all-gather of every block - what multifm_amd runs); 'torch' = one process "
                         "per GPU with torch.distributed collectives (rounds 1-4)")
    ap.add_argument("--group-shards", type=int, default=0,
                    help="TEST AID: run the group path with this many shards on device 0 (MFM_F_GROUP_SHARED_DEVICE; needs a transport "
                         "that accepts one device several times: tests/hoststub/fake_rccl.cpp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the float32-IQ comparison line")
    ap.add_argument("--no-chain", action="store_true", help="skip the device-resident FLEX chain line")
    ap.add_argument("--no-series", action="store_true", help="skip the block-size series and the host-fed end-to-end line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


def _native_oracle(ora):
    """The oracle's C restatement rebuilt on this host with the reference's own optimisation flags (-O2 -march=native,
    CMakeLists.txt:77); the shipped liboracle.so is built for x86-64-v3 because it has to run wherever the tests go.
    Returns (run_channels callable, description)."""
    import ctypes as C
    import subprocess
    import tempfile
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
    srcs = [os.path.join(odir, f) for f in ("mfm_oracle.c", "pocsag_oracle.c", "f32_oracle.c")]
    cmd = ["gcc", "-std=gnu11", "-O2", "-march=native", "-ffp-contract=off", "-fwrapv", "-fPIC", "-D_GNU_SOURCE", "-shared",
           "-o", out] + srcs + ["-lm", "-lpthread"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
        lib = C.CDLL(out)
        i16p = C.POINTER(C.c_int16)
        lib.mfmo_run_channels.argtypes = [i16p, C.c_size_t, C.c_size_t, i16p, i16p, C.c_size_t, C.c_uint, i16p, i16p, i16p,
                                          C.c_size_t, C.c_uint]
        lib.mfmo_run_channels.restype = C.c_size_t
    except Exception:
        return (lambda iq, cre, cim, incr, decim, threads: ora.run_channels(iq, cre, cim, incr, decim, threads=threads)[0],
                "oracle/liboracle.so as shipped (-O2 -march=x86-64-v3; the native rebuild failed)")

    def run(iq, cre, cim, incr, decim, threads):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        nch, T = cre.shape
        nout = (iq.shape[0] - T) // decim + 1
        pcm = np.zeros((nch, nout), np.int16)
        p = lambda a: a.ctypes.data_as(i16p)  # noqa: E731
        got = lib.mfmo_run_channels(p(iq), iq.shape[0], nch, p(cre), p(cim), T, decim, p(incr), p(pcm), None, nout, threads)
        assert got == nout
        return pcm

    return run, "oracle/*.c rebuilt here with gcc -O2 -march=native (the reference's flags, CMakeLists.txt:77)"


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def verify_last_block(pkg, eng, fs, decim, taps, offs, gains, outputs_before):
    """Self-check of the measured shape (VERDICT r02 'Next round' 2b): the PCM the LAST timed launch left in HBM against the
    oracle, for four channels spread over the rows x three windows of 1024 outputs (start, middle, end of the block).
    The oracle is the checker here, never the thing measured: this runs after the timed region.  What the last launch read
    ([history tail | blocks], int16 pairs) is fetched from where the engine says it lies (mfm_engine_last_launch_input),
    `outputs_before` = outputs per channel the stream had produced before that launch (the rotator of output n has been
    stepped n times, filter/direct_fir.c:166-167)."""
    import ctypes as C
    from __graft_entry__ import load_oracle
    ora = load_oracle()
    dptr, stride, nout, _ = eng.last_output_device()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    T = len(taps)
    iptr, n_avail, fmt = eng.last_launch_input()
    assert nout == (n_avail - T) // decim + 1, (nout, n_avail, fmt)
    if fmt == 0:
        host_in = np.empty((n_avail, 2), np.int16)
        if hip.hipMemcpy(host_in.ctypes.data, C.c_void_p(iptr), 4 * n_avail, 2) != 0:
            raise SystemExit("hipMemcpy of the last launch's input failed")
    else:
        raw = np.empty((n_avail, 2), np.uint8)
        if hip.hipMemcpy(raw.ctypes.data, C.c_void_p(iptr), 2 * n_avail, 2) != 0:
            raise SystemExit("hipMemcpy of the last launch's input failed")
        host_in = ora.unpack_bytes(raw, fmt).reshape(-1, 2)  # widened as the reference's front end widens it on the host
    chans = sorted(set(int(round(k * (len(offs) - 1) / 3)) for k in range(4)))
    wins = sorted(set([1, max(1, nout // 2 - 512), max(1, nout - 1024)]))
    bad, checked = 0, 0
    for c in chans:
        row = np.empty(nout, np.int16)
        rc = hip.hipMemcpy(row.ctypes.data, C.c_void_p(dptr + 2 * stride * c), 2 * nout, 2)
        if rc != 0:
            raise SystemExit(f"hipMemcpy of the PCM row failed: {rc}")
        cre, cim, incr = eng.get_channel(c)
        for w0 in wins:
            cnt = min(1024, nout - w0)
            want = ora.window_pcm(host_in, cre, cim, decim, incr, outputs_before, w0, cnt)
            bad += int((row[w0: w0 + cnt] != want).sum())
            checked += cnt
    return {"verified": bad == 0, "mismatches": bad, "outputs_checked": checked, "channels": chans, "windows": wins,
            "outputs_before_block": int(outputs_before),
            "how": "last timed launch's PCM in HBM vs oracle/mfm_oracle.c on the same input window, bit-exact"}


def cpu_baseline(pkg, fs, decim, taps, offs, gains, target_s):
    """The oracle (a port of the reference's per-channel loop) on the host cores, thread-per-channel
    like multifm/receiver.c:89-95, on a bounded sample of the same workload."""
    from __graft_entry__ import load_oracle
    ora = load_oracle()
    run, how = _native_oracle(ora)
    cores = os.cpu_count() or 1
    nch = len(offs)
    cre = np.ascontiguousarray(np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)]), np.int16)
    cim = np.ascontiguousarray(np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)]), np.int16)
    incr = np.ascontiguousarray(np.stack([ora.rot_incr(int(o), fs, decim) for o in offs]), np.int16)
    threads = min(cores, nch)
    n_cal = 1 << 19
    iq = pkg.synth.synth_iq(n_cal, fs, offs[:4], seed=7)
    # one channel on one core: the figure SURVEY.md section 6 quotes for the reference's scalar path (434 MSamp/s)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < min(2.0, target_s / 4):
        run(iq, cre[:1], cim[:1], incr[:1], decim, 1)
        reps += 1
    one_core = reps * n_cal / (time.perf_counter() - t0) / 1e6
    t0 = time.perf_counter()
    run(iq, cre, cim, incr, decim, threads)
    t_cal = time.perf_counter() - t0
    n = int(min(1 << 24, max(n_cal, n_cal * target_s / max(t_cal, 1e-3))))
    big = np.tile(iq, (-(-n // n_cal), 1))[:n]
    run(big, cre, cim, incr, decim, threads)  # first pass only warms the pages and the thread pool
    passes = 0
    t0 = time.perf_counter()
    while True:
        run(big, cre, cim, incr, decim, threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= target_s:
            break
    out = {"value": passes * n * nch / dt / 1e6, "unit": "MSamp/s x channels", "cores": threads, "kind": "port",
           "host_cores": cores, "host_cpu": _cpu_model(),
           "msamp_per_s_one_channel_one_core": one_core,
           "sample": f"{passes} passes over {n} IQ samples x {nch} channels, {how}, {threads} threads "
                     f"thread-per-channel, {dt:.1f} s"}
    if cores > nch:
        # SURVEY.md 8(d)(ii) asks for all cores; thread-per-channel cannot use more threads than channels, so the all-cores
        # figure is taken on a plan with at least as many channels as the host has cores (same filter, same decimation)
        nch2 = max(256, cores)
        fs2, decim2, taps2, offs2, gains2 = pkg.synth.plan("cfg3_1024ch", nr_channels=nch2)
        cre2 = np.ascontiguousarray(np.stack([ora.make_taps(taps2, int(o), fs2, float(g))[0] for o, g in zip(offs2, gains2)]), np.int16)
        cim2 = np.ascontiguousarray(np.stack([ora.make_taps(taps2, int(o), fs2, float(g))[1] for o, g in zip(offs2, gains2)]), np.int16)
        incr2 = np.ascontiguousarray(np.stack([ora.rot_incr(int(o), fs2, decim2) for o in offs2]), np.int16)
        n2 = max(n_cal, n // 4)
        run(big[:n2], cre2, cim2, incr2, decim2, cores)
        p2, t0 = 0, time.perf_counter()
        while True:
            run(big[:n2], cre2, cim2, incr2, decim2, cores)
            p2 += 1
            dt2 = time.perf_counter() - t0
            if dt2 >= target_s / 2:
                break
        out["all_cores"] = {"value": p2 * n2 * nch2 / dt2 / 1e6, "unit": "MSamp/s x channels", "cores": cores, "channels": nch2,
                            "sample": f"{p2} passes over {n2} IQ samples x {nch2} channels (cfg3_1024ch plan), {cores} threads, {dt2:.1f} s"}
    return out


def flex_chain(pkg, torch, fs, decim, taps, offs, gains, block, iters=12):
    """SURVEY.md 8f rows behind the headline kernel, on the same block and channels, nothing leaving HBM: channel engine ->
    16/25 resampler with 821 taps (what the reference's decoder runs in front of FLEX) -> FLEX stage; per-stage times from
    events on the engine's stream.  Outside the timed region, never part of `value`."""
    lib = pkg.load_library()
    in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
    bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = pkg.Engine(fs, decim, block, device=torch.cuda.current_device(), flags=pkg.binding.MFM_F_DEVICE_ONLY,
                     ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7)
    host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
    for b in bufs:
        b.copy_(torch.from_numpy(host))
    rt = pkg.synth.design_lpf(821, 0.45 / 25, 1.0) * 16
    rs = pkg.Resampler(len(offs), np.array([int(t * 16384.0) for t in rt], dtype=np.int16), 16, 25, block // decim + 8,
                       device=torch.cuda.current_device())
    fx = pkg.Flex(len(offs), rs.max_out(), device=torch.cuda.current_device())
    st = torch.cuda.ExternalStream(eng.stream)
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(iters)]
    ny = 0
    for i in range(-40, iters):
        eng.acquire_input()
        if i >= 0:
            marks[i][0].record(st)
        eng.submit(block, producer_stream=0, wait_producer=False)
        dptr, stride, nout, _ = eng.last_output_device()
        if i >= 0:
            marks[i][1].record(st)
        yptr, ystride, ny = rs.process_device(dptr, stride, nout, stream=eng.stream)
        if i >= 0:
            marks[i][2].record(st)
        fx.process_device(yptr, ystride, ny, stream=eng.stream)
        if i >= 0:
            marks[i][3].record(st)
    eng.sync()
    torch.cuda.synchronize()
    t = np.array([[m[0].elapsed_time(m[k]) for k in (1, 2, 3)] for m in marks])
    med = np.median(np.diff(np.concatenate([np.zeros((iters, 1)), t], 1), axis=1), 0)
    total = float(np.median(t[:, 2]))
    for o in (eng, rs, fx):
        o.close()
    return {"chain": "engine -> resampler 16/25 (821 taps) -> FLEX stage, device resident", "block_samples": block,
            "pcm_16k_per_channel": int(ny), "ms_engine": float(med[0]), "ms_resampler": float(med[1]), "ms_flex": float(med[2]),
            "ms_per_block": total, "value": block * len(offs) / total / 1e3, "unit": "MSamp/s x channels"}


def sparse_timing(steps):
    """HIP event pairs on one launch in four (True) or on every launch (False; BENCH_EVENTS=all).  An event pair is two
    command-processor packets between two kernels: on every launch of the driver's 20-step run they cost 2 % of `value`
    (profiles/r05_step_overheads.txt: ms_per_step 0.1183 against 0.1160 over four alternating runs each), so the events stay
    sparse and EVERY launch is timed by the kernel's own 100 MHz stamps instead (`roofline.clocks.kernel_ms_by_stamps`)."""
    return steps >= 8 and os.environ.get("BENCH_EVENTS", "sparse") != "all"   # (a handful of steps: every launch, or none would be timed)


def library_sha16(pkg):
    """first 16 hex digits of the SHA-256 over the channel kernels' sources (tsl-sdr_amd/csrc/mfm_kernel*, mfm_v3_device.h,
    mfm_numerics.h, the Makefile - not the engine: which instance ran is in instance_name()) and
    the ROCm release: ties the instruction counts in profiles/ to the code they belong to.  (Not the .so's own hash: the
    fat binary embeds the build directory, the same sources built elsewhere hash differently - tried.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "tsl-sdr_amd")
    try:
        kernel_files = [f for f in glob.glob(os.path.join(base, "csrc", "*"))
                        if os.path.basename(f).startswith("mfm_kernel") or os.path.basename(f) in ("mfm_v3_device.h", "mfm_numerics.h")]
        for f in sorted(kernel_files + [os.path.join(base, "Makefile")]):
            if os.path.isfile(f):
                h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
        try:   # (a file, not `hipcc --version`: no child process is started from a process that has touched the GPU)
            ver = open("/opt/rocm/.info/version").read().strip()
        except OSError:
            ver = "?"
        h.update(ver.encode())
        return h.hexdigest()[:16]
    except OSError:
        return None


def instance_name(pkg, st, in8):
    """key of a (kernel, geometry) in profiles/r*_issue_model.json"""
    return "variant%d_ch%d_taps%d_ksteps%d_mask%x_tile%d%s%s" % (st["kernel_variant"], st["nr_channels"], st["nr_taps"], st["k_steps"],
                                                                  st["tap_hi_mask"], st["outputs_per_tile"], "_in8" if in8 else "",
                                                                  "_slice128" if st.get("slice_channels") == 128 else "")


BOARD_SAMPLE_AFTER_S = float(os.environ.get("BENCH_BOARD_SAMPLE_AFTER_S", "0.6"))


def device_pci_address(device_index=0):
    """'dddd:bb:dd.f' of a HIP device as sysfs spells it, or None (torch is plumbing here: it asks the runtime)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None


def gpu_sysfs_sample(local_rank=0, pci_address=None, sysfs="/sys"):
    """One sample of the board's shader clock and power from sysfs (what rocm-smi prints, without starting a process inside the
    timed region): {sclk_mhz, power_w, source} or None where the files are not there or not readable.  The card is the one at
    `pci_address` (a box shows the hwmon files of every GPU of its node, usable or not: "the local_rank-th card" read an idle
    neighbour on such a box - round 5); without an address, or without a match, the local_rank-th card and `matched` false."""
    import glob
    cards = []
    numbered = [d for d in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device")) if os.path.basename(os.path.dirname(d))[4:].isdigit()]
    for dev in sorted(numbered, key=lambda d: int(os.path.basename(os.path.dirname(d))[4:])):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw:
            cards.append((os.path.basename(os.path.realpath(dev)).lower(), hw[0]))
    if not cards:
        return None
    by_addr = [h for a, h in cards if pci_address and a == pci_address.lower()]
    hw = by_addr[0] if by_addr else cards[min(local_rank, len(cards) - 1)][1]

    def rd(name):
        try:
            return float(open(os.path.join(hw, name)).read().strip())
        except (OSError, ValueError):
            return None
    sclk = rd("freq1_input")
    power = rd("power1_average") or rd("power1_input")
    cap = rd("power1_cap")
    if sclk is None and power is None:
        return None
    return {"sclk_mhz": sclk / 1e6 if sclk else None, "power_w": power / 1e6 if power else None,
            "power_cap_w": cap / 1e6 if cap else None, "power_of_cap": (power / cap) if (power and cap) else None, "source": hw, "pci_address": pci_address, "matched": bool(by_addr),
            "cards_visible": len(cards)}


def energy_figures(smi, ms_per_step, channels, outputs_per_launch):
    """What a step costs in energy, from the sustained board sample: PPT x step time.  At the board's power cap this - not issue
    slots - is what sets the step time (profiles/r05_exp_issue_vs_power.txt, r05_power_trace.txt)."""
    if not smi or not smi.get("power_w") or not smi.get("matched"):
        return None
    j = smi["power_w"] * ms_per_step * 1e-3
    return {"joule_per_step": j, "nJ_per_channel_output": j * 1e9 / max(1.0, channels * outputs_per_launch),
            "power_w": smi["power_w"], "power_of_cap": smi.get("power_of_cap"),
            "basis": "board_sample.power_w (PPT of the run's own card, 0.6 s into sustained load) x ms_per_step"}


class BoardSampler:
    """One gpu_sysfs_sample() taken WHILE the timed steps run, on a thread of its own: a hwmon read is a message to the SMU and
    takes a millisecond or two - on the submitting thread it would sit in the middle of a 2.4 ms timed region (first try of this
    round: ms_per_step 0.124 -> 0.221).  start() right before the timed loop, result() behind it."""

    def __init__(self, local_rank=0):
        import threading
        self.out, self.t_rel = None, None
        self.local_rank = local_rank
        self.pci_address = device_pci_address(local_rank)   # (asked here, on the caller's thread, not inside the sample)
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        t0 = time.perf_counter()
        self.out = gpu_sysfs_sample(self.local_rank, self.pci_address)
        if self.out is not None:
            self.out["read_ms"] = (time.perf_counter() - t0) * 1e3

    def start(self):
        self.thread.start()

    def result(self):
        self.thread.join(timeout=5.0)
        return self.out


def launch_clocks(eng, n):
    """The last n launches as the kernel stamped them itself (mfm_engine_get_launch_cycles): median shader-clock ticks of the
    longest workgroup, the 100 MHz reference ticks beside them, and the clock that makes of the two."""
    sh, ref = eng.launch_cycles(int(n))
    ok = (sh > 0) & (ref > 0)
    if not ok.any():
        return None
    sh, ref = sh[ok].astype(np.float64), ref[ok].astype(np.float64)
    return {"launches": int(ok.sum()), "shader_ticks_median": float(np.median(sh)), "ref_ticks_100mhz_median": float(np.median(ref)),
            "kernel_ms_by_ref_ticks": float(np.median(ref)) / 1e5,
            # every launch of the timed region, first workgroup's start to last workgroup's end (no dispatch latency in it)
            "kernel_ms_by_stamps": {"launches": int(ok.sum()), "mean": float(np.mean(ref)) / 1e5, "min": float(np.min(ref)) / 1e5,
                                    "median": float(np.median(ref)) / 1e5, "max": float(np.max(ref)) / 1e5},
            "sclk_mhz_effective": float(np.median(sh / ref)) * 100.0,
            "source": "s_memtime / s_memrealtime stamped by the kernel's workgroups, this run"}


def issue_model(kname_full, st, cycles, hbm_frac, lib_sha16):
    """How much of the launch the SIMDs spent issuing: (matrix instructions x 16 + other vector instructions x 3 cycles) / 1024
    SIMDs against THIS run's shader cycles.  The instruction counts are properties of (binary, geometry), not of a box: SQ_INSTS_*
    per launch from the committed rocprofv3 passes of this command (profiles/r*_issue_model.json), only used when that file was
    collected on this very library (sha) and kernel instance.  ceiling_frac = the HBM fraction the kernel would reach if every
    issue slot of the launch were used."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_model.json")), reverse=True)   # the newest round's first
    if cycles is None or not paths:
        return None
    im, ent, path = None, None, paths[0]
    for path in paths:
        im = json.load(open(path))
        ent = im.get("shapes", {}).get(kname_full) if im.get("library_sha16") == lib_sha16 else None
        if ent is not None:
            break
    if ent is None:
        return {"ceiling_frac": None, "reason": "no profiles/r*_issue_model.json was collected on this build and kernel instance",
                "library_sha16": lib_sha16, "profile_library_sha16": im.get("library_sha16")}
    mfma, valu = ent["mfma_insts_per_launch"], ent["other_valu_insts_per_launch"]
    issue_cycles = (mfma * MFMA_ISSUE_CYCLES + valu * VALU_ISSUE_CYCLES) / SIMDS
    busy = issue_cycles / cycles["shader_ticks_median"]
    return {"mfma_insts_per_launch": mfma, "other_valu_insts_per_launch": valu,
            "counts_source": "profiles/" + os.path.basename(path) + ": SQ_INSTS_MFMA, SQ_INSTS_VALU - SQ_INSTS_MFMA per launch (rocprofv3 --pmc of this "
                             "command on this library; instruction counts do not depend on the box)",
            "mfma_issue_cycles": MFMA_ISSUE_CYCLES, "valu_issue_cycles": VALU_ISSUE_CYCLES, "simds": SIMDS,
            "issue_cycles_per_simd": issue_cycles, "launch_shader_cycles": cycles["shader_ticks_median"],
            "simd_busy_fraction": busy, "ceiling_frac": hbm_frac / busy if busy > 0 else None,
            "library_sha16": lib_sha16}


def north_star_shape(pkg, torch, block, steps=12, settle_s=0.3):
    """north_star's target shape in the driver's line (outside the timed region, never part of `value`): the 1024 channels of
    the configs[2] plan on ONE GPU (the reference builds one demod_thread per "channels" entry without limit,
    multifm/receiver.c:195-244), 2^26-sample blocks resident in HBM, same protocol as the headline.  The kernel is
    compute-bound there: `bound_frac` is the HBM fraction the matrix instructions it has to issue would allow all by
    themselves - what `frac` is to be read against, not 0.70."""
    b = pkg.binding
    lib = pkg.load_library()
    try:
        nch = 1024
        fs, decim, taps, offs, gains = pkg.synth.plan("cfg3_1024ch", nr_channels=nch)
        in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
        bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
        eng = pkg.Engine(fs, decim, block, device=torch.cuda.current_device(), flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING,
                         ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
        for o, g in zip(offs, gains):
            eng.add_channel(int(o), taps, float(g))
        eng.commit()
        base = pkg.synth.synth_iq(1 << 22, fs, offs[:: len(offs) // 8][:8], seed=13)
        host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
        for t in bufs:
            t.copy_(torch.from_numpy(host))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < settle_s:
            for _ in range(4):
                eng.acquire_input()
                eng.submit(block, producer_stream=0, wait_producer=False)
            eng.sync()
        w0 = time.perf_counter()
        for _ in range(steps):
            eng.acquire_input()
            eng.submit(block, producer_stream=0, wait_producer=False)
        eng.sync()
        wall_ms = (time.perf_counter() - w0) * 1e3 / steps
        per = eng.launch_ms(steps).astype(np.float64)
        ms = float(np.mean(per))
        cyc = launch_clocks(eng, steps)
        st = eng.stats()
        eng.close()
        del bufs
        T, n_out = len(taps), block // decim
        bytes_alg = 4.0 * block + 2.0 * nch * n_out                       # 4 + 2 C / D bytes per input sample
        ops4 = 2.0 * 4.0 * 4.0 * nch * T * n_out                          # four byte-plane products of 4 real MACs per complex tap
        ks, hi = max(1, st["k_steps"]), bin(st["tap_hi_mask"]).count("1")
        issued = ops4 * (2.0 * ks + 2.0 * hi) / (4.0 * ks)
        frac = bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        out = {"workload": "cfg3_1024ch: 1024 FM channels on one GPU, 128-tap 25 kHz LPF, decimation 96, 2.4 MS/s-shaped int16 IQ, "
                           "block 2^26 samples", "channels": nch, "block_samples": block,
               "value": block * nch / ms / 1e3, "unit": "MSamp/s x channels", "kernel_ms": ms,
               "kernel_ms_min": float(per.min()), "kernel_ms_max": float(per.max()), "wall_ms_per_block": wall_ms,
               "kernel_variant": st["kernel_variant"], "rot_exact_channels": st["rot_exact_channels"],
               "slice_channels": st.get("slice_channels"),   # 128: the long-filter kernel's two-row-block form (from 512 channels on)
               "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": frac, "bytes_per_launch": bytes_alg},
               "compute_roofline": {
                   "bound": "mfma_i8", "unit": "TOP/s", "four_plane_ops_per_launch": ops4, "issued_ops_per_launch": issued,
                   "peak_nominal": MFMA_I8_PEAK_TOPS, "peak_measured_guide": MFMA_I8_MEASURED_TOPS,
                   "frac_four_planes_of_nominal": ops4 / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                   "frac_issued_of_nominal": issued / (ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                   "frac_issued_of_measured": issued / (ms * 1e-3) / 1e12 / MFMA_I8_MEASURED_TOPS,
                   "mfma_only_ms_at_nominal": issued / (MFMA_I8_PEAK_TOPS * 1e12) * 1e3,
                   "mfma_only_ms_at_measured": issued / (MFMA_I8_MEASURED_TOPS * 1e12) * 1e3},
               # the HBM fraction a kernel of nothing but its matrix instructions would show: what `frac` stands against
               "bound_frac": {"at_nominal_5000_tops": bytes_alg / (issued / (MFMA_I8_PEAK_TOPS * 1e12)) / 1e9 / HBM_PEAK_GBPS,
                              "at_measured_3944_tops": bytes_alg / (issued / (MFMA_I8_MEASURED_TOPS * 1e12)) / 1e9 / HBM_PEAK_GBPS},
               "clocks": cyc}
        out["frac_of_bound"] = {k: frac / v for k, v in out["bound_frac"].items()}
        # the same issue model as the headline's (matrix + other vector instructions against this run's shader cycles)
        out["instance"] = instance_name(pkg, st, False)
        out["issue_model"] = issue_model(out["instance"], st, cyc, frac, library_sha16(pkg))
        return out
    except Exception as e:  # a side line must never take the headline down
        return {"error": repr(e)}


def other_geometries(pkg, torch, block, steps=24, settle_s=0.25):
    """The reference's other deployed geometries on the same engine, outside the timed region (never part of `value`): the
    int16 path of BASELINE configs[4] (per-GPU share: 256 of the 2048 Airspy channels, D = 400, 512-tap low-pass - the
    resident long-filter instances), the channelizer geometry of configs[3] (etc/pocsag_rtlsdr.json:
    1.2 MS/s, D = 25) and of etc/multifm.json (1 MS/s, D = 40), and the three front-end configurations with the longer low-pass
    files the reference ships for their sample rates (profiles/r04_etc_shapes.txt), 64 channels each.  Same protocol as the headline: blocks
    resident in HBM, a settle phase of back-to-back launches, kernel duration from the engine's HIP events."""
    b = pkg.binding
    lib = pkg.load_library()
    out = {}
    for key, plan, nch in (("configs4_int16_share", "cfg5_airspy", 256), ("configs3_pocsag_d25", "pocsag_rtlsdr", 64),
                           ("multifm_json_d40", "multifm_1ch", 64),
                           # the same front ends with the low-pass files the reference ships for their sample rates
                           ("pocsag_rtlsdr_d25_256taps", "pocsag_rtlsdr_256taps", 64), ("pocsag_airspy_d100_256taps", "pocsag_airspy", 64),
                           ("multifm_airspy_d120_512taps", "multifm_airspy", 64)):
        try:
            fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=nch)
            in_bytes = lib.mfm_engine_input_bytes(block, len(taps))
            bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
            eng = pkg.Engine(fs, decim, block, device=torch.cuda.current_device(), flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING,
                             ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
            for o, g in zip(offs, gains):
                eng.add_channel(int(o), taps, float(g))
            eng.commit()
            base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=11)
            host = np.tile(base, (-(-(in_bytes // 4) // base.shape[0]), 1))[: in_bytes // 4].reshape(-1)
            for t in bufs:
                t.copy_(torch.from_numpy(host))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < settle_s:
                for _ in range(8):
                    eng.acquire_input()
                    eng.submit(block, producer_stream=0, wait_producer=False)
                eng.sync()
            for _ in range(steps):
                eng.acquire_input()
                eng.submit(block, producer_stream=0, wait_producer=False)
            eng.sync()
            ms = float(np.mean(eng.launch_ms(steps)))
            st = eng.stats()
            eng.close()
            del bufs
            out[key] = {"channels": nch, "sample_rate_hz": fs, "decimation": decim, "taps": len(taps), "block_samples": block,
                        "kernel_ms": ms, "value": block * nch / ms / 1e3, "unit": "MSamp/s x channels",
                        "kernel_variant": st["kernel_variant"], "taps_resident": st["taps_resident"],
                        "k_steps": st["k_steps"], "tap_hi_mask": st["tap_hi_mask"]}
        except Exception as e:  # a side line must never take the headline down
            out[key] = {"error": repr(e)}
    return out


def block_series(pkg, torch, fs, decim, taps, offs, gains, total_log2=33, settle_s=0.3):
    """SURVEY.md 8(d)'s protocol - blocks of 2^20 samples (4 MiB; >= 64 per timing), and 2^22 and 2^26 beside them - resident in
    HBM and handed to the engine ONE BY ONE by a C producer loop (mfm_engine_replay: acquire_input + submit per block, what
    a C host does per delivered sample_buf).  Engines per block size: `coalesced` gathers the backlog into launches of up
    to 2^26 samples (mfm_engine_config::coalesce_samples; a channel thread of the reference drains up to 128 queued
    buffers in one go, multifm/demod.c:134-150,297), `per_block` launches every block on its own (coalesce_samples = 0);
    both with consecutive launches on two streams (MFM_F_OVERLAP), and `_one_stream` without.
    Wall time around the loop + sync; frac = SURVEY 8(d)'s algorithmic bytes / that time / 8 TB/s.  Outside the timed region,
    never part of `value`."""
    b = pkg.binding
    out = {"protocol": "blocks resident in HBM, C producer loop (mfm_engine_replay), wall clock around loop + sync after a "
                       f"settle phase; 2^{total_log2} samples per timing", "coalesce_samples": 1 << 26, "series": []}
    nch, T = len(offs), len(taps)
    for blog in (20, 22, 26):
        block = 1 << blog
        row = {"block_samples": block, "blocks": (1 << total_log2) // block}
        for mode, co, fl in (("coalesced", 1 << 26, b.MFM_F_OVERLAP), ("per_block", 0, b.MFM_F_OVERLAP),
                             ("coalesced_one_stream", 1 << 26, 0), ("per_block_one_stream", 0, 0)):
            try:
                eng = pkg.Engine(fs, decim, block, device=torch.cuda.current_device(), flags=b.MFM_F_DEVICE_ONLY | fl,
                                 coalesce_samples=co)
                for o, g in zip(offs, gains):
                    eng.add_channel(int(o), taps, float(g))
                eng.commit()
                cfg = b.EngineConfig()
                cfg.max_block_samples, cfg.coalesce_samples = block, co
                in_bytes = eng.lib.mfm_engine_input_bytes_cfg(ctypes_byref(cfg), T, None)
                base = torch.from_numpy(pkg.synth.synth_iq(1 << 20, fs, offs[:: max(1, nch // 8)][:8], seed=7).reshape(-1)).cuda()
                seen = set()
                for _ in range(3):
                    ptr, cap = eng.acquire_input()
                    if ptr not in seen:   # fill every input buffer once (they are the engine's own)
                        seen.add(ptr)
                        n16 = (in_bytes - 4 * (2 * T + 64)) // 2
                        dst = _as_tensor(torch, ptr, n16)
                        reps = -(-n16 // base.numel())
                        dst.copy_(base.repeat(reps)[:n16])
                    torch.cuda.synchronize()
                    eng.submit(block, producer_stream=0, wait_producer=False)
                    eng.flush()
                eng.sync()
                nblk = row["blocks"]
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < settle_s:
                    eng.replay(block, max(1, nblk // 8))
                eng.sync()
                st0 = eng.stats()
                t0 = time.perf_counter()
                eng.replay(block, nblk)
                t_sub = time.perf_counter() - t0
                eng.sync()
                dt = time.perf_counter() - t0
                st1 = eng.stats()
                eng.close()
                outs = st1["outputs"] - st0["outputs"]
                alg = nblk * block * 4 + nch * outs * 2
                row[mode] = {"launches": st1["launches"] - st0["launches"], "us_per_block": dt / nblk * 1e6,
                             "producer_us_per_block": t_sub / nblk * 1e6,
                             "input_msamp_per_s": nblk * block / dt / 1e6, "value": nblk * block * nch / dt / 1e6,
                             "achieved_GBps": alg / dt / 1e9, "frac": alg / dt / 1e9 / HBM_PEAK_GBPS}
            except Exception as e:  # a side line must never take the headline down
                row[mode] = {"error": repr(e)}
        out["series"].append(row)
    return out


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


def _as_tensor(torch, ptr, n_int16):
    """a torch view of n_int16 int16 elements of device memory at `ptr` (the engine's own input buffer)"""
    class _Ext:
        pass
    holder = _Ext()
    holder.__cuda_array_interface__ = {"shape": (n_int16,), "typestr": "<i2", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(holder, device="cuda")


def end_to_end(pkg, fs, decim, taps, offs, gains, buf_samples=131072, nr_bufs=4096):
    """"a second end-to-end number includes H2D from pinned host memory".  What multifm_amd does per delivered
    sample_buf, in the size the RTL-SDR front end delivers (131 072 samples, multifm/rtl_sdr_if.c:46): host buffer ->
    mfm_group_push (staging in pinned memory, H2D, one device group of one GPU) -> kernel -> PCM mirrored to pinned host
    memory -> mfm_group_fetch / release, with the group gathering up to a pool's worth of buffers (128, demod.c:297) per
    launch as the C host configures it.  `pinned_pool`: the buffers are page-locked like the C host's pool and the H2D reads
    them in place (mfm_group_push_pinned); `staged_copy`: pageable buffers through the engine's staging copy.  `c_loop`: the
    producer loop in C (mfm_group_replay_pinned), otherwise a Python loop (one ctypes call per buffer).  PCIe inclusive; never
    `value`."""
    b = pkg.binding
    out = {"buffer_samples": buf_samples, "buffers": nr_bufs,
           "path": "host buffer -> mfm_group_push -> H2D -> kernel -> D2H -> mfm_group_fetch/release"}
    nch = len(offs)
    data = pkg.synth.synth_iq(buf_samples * 8, fs, offs[:: max(1, nch // 8)][:8], seed=7).reshape(8, -1, 2)
    import ctypes
    lib0 = pkg.load_library()
    # the C host's sample_buf pool is page-locked memory (mfm_host_alloc): eight such buffers stand in for it
    pinned = [lib0.mfm_host_alloc(buf_samples * 4) for _ in range(8)]
    for k, ptr in enumerate(pinned):
        if not ptr:
            return {"error": "mfm_host_alloc failed"}
        ctypes.memmove(ptr, data[k].ctypes.data, buf_samples * 4)
    arr = (ctypes.c_void_p * 8)(*pinned)
    for mode, co, pin in (("c_loop_pinned_pool_coalesced_128_buffers", 128 * buf_samples, "c"),
                          ("c_loop_pinned_pool_launch_per_buffer", 0, "c"),
                          ("pinned_pool_coalesced_128_buffers", 128 * buf_samples, True),
                          ("pinned_pool_launch_per_buffer", 0, True), ("staged_copy_coalesced_128_buffers", 128 * buf_samples, False)):
        try:
            grp = b.Group(fs, decim, buf_samples, devices=(0,), coalesce_samples=co)
            for o, g in zip(offs, gains):
                grp.add_channel(int(o), taps, float(g))
            grp.commit()
            blks = (b.Block * 1)()
            lib = grp.lib

            def drain_one():
                rc = lib.mfm_group_fetch(grp.h, blks)
                if rc == b.MFM_E_DONE:
                    return 0
                if rc != 0:
                    raise RuntimeError(f"mfm_group_fetch: {rc}")
                n = blks[0].nr_outputs
                lib.mfm_group_release(grp.h)
                return n

            def push(i):
                if pin:
                    return lib.mfm_group_push_pinned(grp.h, pinned[i & 7], buf_samples, b.MFM_IN_CS16, None)
                return grp.push(data[i & 7])

            def run(nbuf):
                if pin == "c":
                    # the producer loop in C (mfm_group_replay_pinned): push, fetch / release when the rings are full, flush + drain
                    got = ctypes.c_uint64()
                    rc = lib.mfm_group_replay_pinned(grp.h, arr, 8, buf_samples, b.MFM_IN_CS16, nbuf, ctypes.byref(got))
                    if rc != 0:
                        raise RuntimeError(f"mfm_group_replay_pinned: {rc} {lib.mfm_last_error()}")
                    return got.value
                outs = 0
                for i in range(nbuf):
                    while True:
                        rc = push(i)
                        if rc == 0:
                            break
                        if rc != b.MFM_E_BUSY:
                            raise RuntimeError(f"push: {rc} {lib.mfm_last_error()}")
                        outs += drain_one()
                while True:
                    rc = grp.flush()
                    while True:
                        n = drain_one()
                        if not n:
                            break
                        outs += n
                    if rc == 0:
                        break
                grp.sync()
                while True:
                    n = drain_one()
                    if not n:
                        break
                    outs += n
                return outs

            run(max(64, nr_bufs // 8))
            st0 = grp.stats(0)
            t0 = time.perf_counter()
            outs = run(nr_bufs)
            dt = time.perf_counter() - t0
            st1 = grp.stats(0)
            grp.close()
            n_in = nr_bufs * buf_samples
            out[mode] = {"launches": st1["launches"] - st0["launches"], "input_msamp_per_s": n_in / dt / 1e6,
                         "value": n_in * nch / dt / 1e6, "unit": "MSamp/s x channels", "us_per_buffer": dt / nr_bufs * 1e6,
                         "h2d_GBps": n_in * 4 / dt / 1e9, "d2h_GBps": outs * nch * 2 / dt / 1e9,
                         "outputs_per_channel": int(outs)}
        except Exception as e:  # a side line must never take the headline down
            out[mode] = {"error": repr(e)}
    # what the C host's submit thread does with a backlog since round 5: the pool hands its frames out in address order, so
    # buffers delivered one after the other are neighbours in the slab and go to the device as ONE strided copy command per run
    # (mfm_group_push_pinned_run).  An arena laid out like the pool (64 bytes of sample_buf header between the data of two
    # frames), the producer loop in C (mfm_group_replay_arena); RTL-SDR-sized and file_if-sized buffers.
    for bs, nb, runs in ((buf_samples, nr_bufs, (1, 16, 64)), (4096, 32 * nr_bufs, (1, 64))):
        stride = 64 + bs * 4
        frames = 128 if bs == buf_samples else 2048
        arena = lib0.mfm_host_alloc(frames * stride)
        if not arena:
            continue
        for k in range(frames):
            ctypes.memmove(arena + k * stride + 64, data[k & 7].ctypes.data, bs * 4)
        for mr in runs:
            mode = f"c_loop_pool_arena_{bs}_sample_buffers_runs_of_up_to_{mr}"
            try:
                grp = b.Group(fs, decim, bs, devices=(0,), coalesce_samples=128 * buf_samples)
                for o, g in zip(offs, gains):
                    grp.add_channel(int(o), taps, float(g))
                grp.commit()
                got, cmds = ctypes.c_uint64(), ctypes.c_uint64()

                def run_arena(n):
                    rc = grp.lib.mfm_group_replay_arena(grp.h, ctypes.c_void_p(arena + 64), stride, frames, bs, b.MFM_IN_CS16, n, mr,
                                                        ctypes.byref(got), ctypes.byref(cmds))
                    if rc != 0:
                        raise RuntimeError(f"mfm_group_replay_arena: {rc} {grp.lib.mfm_last_error()}")
                run_arena(max(64, nb // 8))
                st0 = grp.stats(0)
                t0 = time.perf_counter()
                run_arena(nb)
                dt = time.perf_counter() - t0
                st1 = grp.stats(0)
                grp.close()
                n_in = nb * bs
                out[mode] = {"launches": st1["launches"] - st0["launches"], "copy_commands": int(cmds.value), "buffers": nb,
                             "input_msamp_per_s": n_in / dt / 1e6, "value": n_in * nch / dt / 1e6, "unit": "MSamp/s x channels",
                             "us_per_buffer": dt / nb * 1e6, "h2d_GBps": n_in * 4 / dt / 1e9,
                             "d2h_GBps": got.value * nch * 2 / dt / 1e9, "outputs_per_channel": int(got.value)}
            except Exception as e:
                out[mode] = {"error": repr(e)}
        lib0.mfm_host_free(arena)
    # the link by itself, for what the figures above are to be read against: bare hipMemcpyAsync out of one page-locked arena
    # in pieces of one sample_buf (512 KiB) and of 64 MiB, with the PCM's share of bytes coming back on a second stream
    try:
        back = 2.0 * nch / decim / 4.0
        link = {"d2h_bytes_per_h2d_byte": back}
        for name, piece in (("pieces_512KiB", buf_samples * 4), ("pieces_64MiB", 64 << 20)):
            h, d = ctypes.c_double(), ctypes.c_double()
            rc = lib0.mfm_link_probe(0, piece, 2 << 30, back, ctypes.byref(h), ctypes.byref(d))
            link[name] = {"h2d_GBps": h.value, "d2h_GBps": d.value} if rc == 0 else {"error": rc}
            h2, d2 = ctypes.c_double(), ctypes.c_double()
            rc = lib0.mfm_link_probe(0, piece, 2 << 30, 0.0, ctypes.byref(h2), ctypes.byref(d2))
            link[name]["h2d_alone_GBps"] = h2.value if rc == 0 else None
        h, d = ctypes.c_double(), ctypes.c_double()
        rc = lib0.mfm_link_probe_runs(0, buf_samples * 4, 64, 16, 2 << 30, back, ctypes.byref(h), ctypes.byref(d))
        link["pieces_512KiB_runs_of_16_strided"] = {"h2d_GBps": h.value, "d2h_GBps": d.value} if rc == 0 else {"error": rc}
        per_buf = out.get("c_loop_pinned_pool_coalesced_128_buffers", {}).get("h2d_GBps")
        runs16 = out.get(f"c_loop_pool_arena_{buf_samples}_sample_buffers_runs_of_up_to_16", {}).get("h2d_GBps")
        if per_buf and "h2d_GBps" in link["pieces_512KiB"]:
            link["end_to_end_one_command_per_buffer_over_link_512KiB"] = per_buf / link["pieces_512KiB"]["h2d_GBps"]
        if runs16 and "h2d_GBps" in link["pieces_64MiB"]:
            link["end_to_end_runs_of_16_over_link_64MiB"] = runs16 / link["pieces_64MiB"]["h2d_GBps"]
            link["end_to_end_runs_of_16_over_link_strided_runs"] = runs16 / max(1e-9, link["pieces_512KiB_runs_of_16_strided"].get("h2d_GBps", 0.0))
        out["link"] = link
    except Exception as e:
        out["link"] = {"error": repr(e)}
    # latency: deliver -> PCM fetched, for a feed paced at the front ends' rates (one 131 072-sample buffer every 54.6 ms at
    # 2.4 MS/s, every 13.1 ms at 10 MS/s), under the three gathering settings.  The device is idle when a paced buffer arrives,
    # so the engine launches it at once whatever it may gather up to - the figure says what that costs.
    try:
        lat = {}
        for name, co in (("launch_per_buffer", 0), ("gather_up_to_one_pool_128_buffers", 128 * buf_samples), ("gather_up_to_2^26_samples", 1 << 26)):
            grp = b.Group(fs, decim, buf_samples, devices=(0,), coalesce_samples=co)
            for o, g in zip(offs, gains):
                grp.add_channel(int(o), taps, float(g))
            grp.commit()
            blks = (b.Block * 1)()
            lib = grp.lib
            for rate in (2400000, 10000000):
                period = buf_samples / rate
                ms = []
                t_next = time.perf_counter()
                for i in range(20):
                    while time.perf_counter() < t_next:
                        pass
                    t0 = time.perf_counter()
                    rc = lib.mfm_group_push_pinned(grp.h, pinned[i & 7], buf_samples, b.MFM_IN_CS16, None)
                    if rc != 0:
                        raise RuntimeError(f"push: {rc}")
                    got = 0
                    while got == 0 and time.perf_counter() - t0 < 2.0:
                        rc = lib.mfm_group_fetch(grp.h, blks)   # waits for a block that is in flight
                        if rc == 0:
                            got = blks[0].nr_outputs
                            lib.mfm_group_release(grp.h)
                        elif rc == b.MFM_E_DONE:
                            lib.mfm_group_flush(grp.h)             # (nothing launched yet: a deferred buffer)
                        else:
                            raise RuntimeError(f"fetch: {rc}")
                    if i >= 4:
                        ms.append((time.perf_counter() - t0) * 1e3)
                    t_next += period
                lat.setdefault(name, {})[f"feed_{rate / 1e6:g}_MSps"] = {"latency_ms_median": float(np.median(ms)), "latency_ms_max": float(np.max(ms)),
                                                                         "buffers": len(ms), "buffer_period_ms": period * 1e3}
            grp.close()
        out["latency"] = lat
    except Exception as e:
        out["latency"] = {"error": repr(e)}
    for ptr in pinned:
        lib0.mfm_host_free(ptr)
    return out


def ingest_8bit(pkg, fs, decim, taps, offs, gains, block, int16_kernel_ms, steps=6, warmup=2):
    """the same channels fed with an RTL-SDR block (8-bit IQ, multifm/rtl_sdr_if.c:146-148) that the
    matrix kernel reads as bytes, outside the timed region; kernel duration from the engine's HIP
    events.  Never part of `value`."""
    b = pkg.binding
    eng = pkg.Engine(fs, decim, block, device=0, flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    base = pkg.synth.synth_iq(1 << 22, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7).reshape(-1, 2)
    u8 = np.clip((base.astype(np.int32) >> 7) + 127, 0, 255).astype(np.uint8)
    data = np.tile(u8, (block // u8.shape[0] + 1, 1))[:block]
    for _ in range(warmup + steps):
        rc = eng.push_bytes(data, b.MFM_IN_RTLSDR_U8)
        if rc != 0:
            raise SystemExit(f"mfm_engine_push_bytes: {rc}")
    eng.sync()
    ms = float(np.mean(eng.launch_ms()[-steps:]))
    st = eng.stats()
    eng.close()
    alg = block * 2 + len(offs) * (block // decim) * 2
    return {"input": "rtl-sdr u8 IQ, read as bytes by the matrix kernel", "block_samples": block, "kernel_ms": ms,
            "launches_8bit": st["launches_8bit"], "launches": st["launches"], "algorithmic_bytes": alg,
            "hbm_GBps": alg / ms / 1e6, "input_msamp_per_s": block / ms / 1e3,
            "time_vs_int16_kernel": ms / int16_kernel_ms}


def fp32_path(pkg, torch, fs, decim, taps, offs, gains, int16_kernel_ms, int16_block, block_log2=26, iters=30):
    """Kernel time of the floating-point IQ path (mfm_f32_*) on blocks of the headline's length resident in HBM (round 3:
    2^26 samples like the integer path it is compared with - on 2^24-sample blocks a workgroup runs 5.4 tiles and the
    kernel's start and end weigh 10 % of the launch, profiles/r03_f32_block_sizes.txt), behind a settle phase like the
    headline's, HIP events on the launch stream."""
    blk = 1 << block_log2
    base = pkg.synth.synth_iq(1 << 20, fs, offs[:: max(1, len(offs) // 8)][:8], seed=7).astype(np.float32)
    d_f = torch.from_numpy(np.tile(base, (blk // base.shape[0], 1)).reshape(-1)).cuda()
    eng = pkg.F32Engine(fs, decim, blk, device=torch.cuda.current_device())
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    st = torch.cuda.current_stream().cuda_stream
    # the same protocol as the headline: back-to-back launches for a while first (the GPU has idled through the CPU baseline,
    # and the board's power management takes ~100 launches to settle), then the timed ones
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < 0.4:
        for _ in range(8):
            b = eng.process_device(d_f.data_ptr(), blk, stream=st)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b = eng.process_device(d_f.data_ptr(), blk, stream=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    C, T = len(offs), len(taps)
    flops = 8.0 * C * T * b.nr_out
    eng.close()
    return {"kernel": "mfm_f32_channel_kernel", "block_samples": blk, "ms_per_block": ms,
            "value": blk * C / ms / 1e3, "unit": "MSamp/s x channels", "dtype": "f32",
            "achieved_tflops": flops / ms / 1e9, "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
            "frac": flops / ms / 1e9 / FP32_VECTOR_PEAK_TFLOPS,
            "time_vs_int16_path": ms / (int16_kernel_ms * blk / int16_block),
            "tolerance": "1e-5 rel vs fp64 restatement (tests/test_f32_path.py)"}


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILDREN of this process (which has not
    touched a GPU; a process that has must never be replaced by another), hand their output through, exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def hip_memcpy_h2d(dst_ptr, arr):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if hip.hipMemcpy(C.c_void_p(dst_ptr), arr.ctypes.data, arr.nbytes, 1) != 0:
        raise SystemExit("hipMemcpy (host to device) failed")


def group_run(pkg, args, devices, shared, fs, decim, taps, offs, gains, block, steps, warmup, settle_s):
    """One pass of the protocol through the library's device group (mfm_group_*; csrc/mfm_group.hip): the channels are cut into
    contiguous shards, one engine per device, every block goes from the root's input buffer to the others by the library's own
    RCCL exchange (scatter + all-gather) and is submitted on every shard - what host/mfm_receiver.c runs for "gpuDevices": [..].
    One process, one host thread.  Blocks are resident in the root's HBM buffers before the timed region."""
    b = pkg.binding
    S = len(devices)
    g = b.Group(fs, decim, block, devices=tuple(devices),
                flags=b.MFM_F_DEVICE_ONLY | b.MFM_F_TIMING | (b.MFM_F_TIMING_SPARSE if sparse_timing(steps) else 0) |
                (b.MFM_F_GROUP_SHARED_DEVICE if shared else 0),
                exchange=b.MFM_X_RCCL_ALLGATHER if S > 1 else b.MFM_X_AUTO)
    for o, gn in zip(offs, gains):
        g.add_channel(int(o), taps, float(gn))
    g.commit()
    base = pkg.synth.synth_iq(1 << 22, fs, list(offs)[:: max(1, len(offs) // 8)][:8], seed=7)
    host = np.ascontiguousarray(np.tile(base, (-(-block // base.shape[0]), 1))[:block])
    filled = []

    def step():
        ptr, cap = g.acquire_input()
        if all(abs(ptr - q) > (1 << 20) for q in filled):   # the root's two input buffers, each filled once (before the timed region)
            hip_memcpy_h2d(ptr, host)
            filled.append(ptr)
        while g.submit(block) == b.MFM_E_BUSY:
            raise SystemExit("mfm_group_submit: busy in device-only mode")

    for _ in range(4):
        step()
    g.sync()
    t_s = time.perf_counter()
    for _ in range(8):
        step()
    g.sync()
    per = max((time.perf_counter() - t_s) / 8, 1e-6)
    for _ in range(int(min(20000, max(0, settle_s / per)))):
        step()
    g.sync()
    for _ in range(warmup):
        step()
    g.sync()
    st0 = [g.stats(s) for s in range(g.nr_shards)]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    g.sync()
    dt = time.perf_counter() - t0
    st1 = [g.stats(s) for s in range(g.nr_shards)]
    eng0 = g.shard_engine(0)
    cycles = launch_clocks(eng0, min(steps, 1024))
    uses, nblk, moved = g.exchange_info()
    shards = []
    for s in range(g.nr_shards):
        lo, n, dev = g.shard_info(s)
        timed = st1[s]["timed_launches"] - st0[s]["timed_launches"]
        shards.append({"shard": s, "device": dev, "first_channel": lo, "channels": n,
                       "kernel_ms": (st1[s]["kernel_ms"] - st0[s]["kernel_ms"]) / max(1, timed), "timed_launches": int(timed),
                       "launches": int(st1[s]["launches"] - st0[s]["launches"]), "outputs": int(st1[s]["outputs"] - st0[s]["outputs"]),
                       "kernel_variant": st1[s]["kernel_variant"], "k_steps": st1[s]["k_steps"], "tap_hi_mask": st1[s]["tap_hi_mask"],
                       "rot_exact_channels": st1[s]["rot_exact_channels"], "pending_blocks": st1[s]["pending_blocks"],
                       "submits": int(st1[s]["submits"] - st0[s]["submits"])})
    # every shard's last block against the oracle, on the input THAT shard's launch read: a wrong scatter or all-gather offset
    # corrupts the input of the non-root shards only
    per_shard = []
    for s in range(g.nr_shards):
        lo_s, n_s, _ = g.shard_info(s)
        eng_s = g.shard_engine(s)
        n_last = eng_s.last_output_device()[2]
        v = verify_last_block(pkg, eng_s, fs, decim, taps, list(offs)[lo_s:lo_s + n_s], list(gains)[lo_s:lo_s + n_s],
                              g.stats(s)["outputs"] - n_last)
        v["shard"] = s
        per_shard.append(v)
    verified = dict(per_shard[0])
    verified["verified"] = all(v["verified"] for v in per_shard)
    verified["shards_verified"] = [bool(v["verified"]) for v in per_shard]
    if not verified["verified"]:
        verified["failed_shards"] = [v for v in per_shard if not v["verified"]]
    detail = [g.exchange_detail(s) for s in range(g.nr_shards)]
    sampler = BoardSampler(0)   # behind the timed region and the self-check, while the same steps run again (see main())
    k, started, t_load = 0, False, time.perf_counter()
    while k < 16 or ((not started or sampler.thread.is_alive()) and k < 40000):
        step()
        k += 1
        if k % 64 == 0:
            g.sync()
        if not started and time.perf_counter() - t_load >= BOARD_SAMPLE_AFTER_S:
            sampler.start()
            started = True
    g.sync()
    smi = sampler.result()
    if smi is not None:
        smi["when"] = "after %.1f s of further back-to-back steps of the same workload behind the timed region" % BOARD_SAMPLE_AFTER_S
    try:
        rccl_file = b.rccl_library() if uses else None
    except Exception as exc:   # (a group that exchanges has loaded it: this cannot fail there)
        rccl_file = "unavailable: %s" % exc
    out = {"dt": dt, "shards": shards, "exchange": {"uses_rccl": bool(uses), "blocks": int(nblk), "bytes_to_other_devices": int(moved),
                                                    "mode": "scatter + all-gather (MFM_X_RCCL_ALLGATHER)" if S > 1 else "none (one device)",
                                                    "rccl_library": rccl_file,
                                                    "rccl_ranks": detail[0]["rccl_ranks"] if uses else 0,
                                                    "per_shard": detail},
           "verified": verified, "clocks": cycles, "board_sample": smi, "st1": st1[0]}
    g.close()
    return out


def group_main(args, pkg, devices, shared, world_for_line):
    """`--exchange group`: the whole line from one process."""
    S = len(devices)
    cpg = args.channels_per_gpu or 64   # the same per-GPU work at every N (weak scaling): BENCH's 64 channels per GPU
    total_ch = cpg * S
    fs, decim, taps, offs, gains = pkg.synth.plan(args.config, nr_channels=total_ch)
    block = 1 << args.block_log2
    T = len(taps)
    r = group_run(pkg, args, devices, shared, fs, decim, taps, offs, gains, block, args.steps, args.warmup, args.settle_seconds)
    dt = r["dt"]
    sh0 = r["shards"][0]
    k_ms = max(sh["kernel_ms"] for sh in r["shards"])
    outs = sh0["outputs"] / max(1, sh0["launches"])
    bytes_per_launch = 4.0 * block + 2.0 * sh0["channels"] * outs   # per GPU: SURVEY.md 8(d), 4 + 2 C_g / D bytes per input sample
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    msamp = args.steps * block / dt / 1e6
    kname = {0: "mfm_channel_kernel", 1: "mfm_channel_kernel_mfma", 2: "mfm_channel_kernel_v3"}[sh0["kernel_variant"]]
    need = 4.0 * block / (k_ms * 1e-3) / 1e9 if S > 1 else 0.0
    single = None
    if S > 1:
        # the same per-GPU shape on ONE device of the same node, same run, same protocol: what "N x one GPU" means for this line
        fs1, decim1, taps1, offs1, gains1 = pkg.synth.plan(args.config, nr_channels=cpg)
        r1 = group_run(pkg, args, devices[:1], False, fs1, decim1, taps1, offs1, gains1, block, args.steps, args.warmup, args.settle_seconds)
        v1 = args.steps * block / r1["dt"] / 1e6 * cpg
        single = {"value": v1, "unit": "MSamp/s x channels", "channels": cpg, "ms_per_step": r1["dt"] / args.steps * 1e3,
                  "kernel_ms": r1["shards"][0]["kernel_ms"], "verified": r1["verified"]["verified"], "device": devices[0],
                  "what": "a one-device group at the same channels_per_gpu, same blocks, same step count, in this run"}
    line = {
        "metric": "input IQ MSamp/s x channels demodulated", "value": msamp * total_ch, "unit": "MSamp/s x channels",
        "n_gpus": world_for_line, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
        "config": {"workload": f"{args.config}: {cpg} FM channels per GPU ({total_ch} total), {T}-tap 25 kHz LPF, decimation {decim}, "
                               f"fs {fs} Hz-shaped int16 IQ, block 2^{args.block_log2} samples",
                   "channels_per_gpu": cpg, "channels_total": total_ch, "block_samples": block, "decimation": int(decim),
                   "taps": int(T), "sample_rate_hz": int(fs), "input_msamp_per_s": msamp,
                   "parallelism": ("1 GPU, through the device group" if S == 1 else
                                   f"channel shards x{S} in ONE process (mfm_group_*), the library's RCCL scatter + all-gather of every IQ block")
                                  + (" [TEST AID: all shards on one device]" if shared else "")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": kname, "kernel_ms": k_ms, "bytes_per_launch": bytes_per_launch,
                     "per": "GPU (the slowest shard's kernel)", "clocks": r["clocks"], "board_sample": r["board_sample"]},
        "cpu_baseline": None,
        "verified": r["verified"]["verified"], "verification": r["verified"],
        "group": {"shards": r["shards"], "exchange_info": r["exchange"],
                  "needed_GBps_per_peer": need, "api": "mfm_group_acquire_input + mfm_group_submit (include/multifm_hip.h)"},
        "single_gpu_same_shape": single,
        "scaling_efficiency": (msamp * total_ch) / (S * single["value"]) if single else None,
        "protocol": {"settle_seconds": args.settle_seconds, "warmup_steps": args.warmup, "timed_steps": args.steps},
    }
    return line


def start_watchdog(args):
    """N > 1 only: a run that does not finish (a collective that never completes on hardware this code has not met) ends with
    a message and a non-zero exit instead of sitting there until the driver's limit.  BENCH_WATCHDOG_S, default 900."""
    if args.gpus <= 1:
        return
    import threading
    limit = float(os.environ.get("BENCH_WATCHDOG_S", "900"))

    def fire():
        sys.stderr.write(f"bench.py: no result after {limit:.0f} s with --gpus {args.gpus} (rank {os.environ.get('RANK', '0')}): giving up\n")
        sys.stderr.flush()
        os._exit(3)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()


def main():
    args = parse()
    start_watchdog(args)
    group = args.exchange == "group" and (args.gpus > 1 or args.group_shards > 0)
    group_error = None
    if group:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        line = None
        if world > 1:
            # the ranks the driver started: rank 0 drives every GPU of the node through the device group, the others wait for its
            # word (they have not touched a GPU).  Should the group not come up on this node - a librccl that does not load, a
            # communicator that cannot be made in one process - rank 0 says so and EVERY rank goes on to the per-rank form below
            # (one engine per rank, torch.distributed = RCCL moves the block): a line with the reason in it instead of none.
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        if rank == 0:
            from __graft_entry__ import load_package
            pkg = load_package()
            shared = args.group_shards > 0
            devices = [0] * args.group_shards if shared else list(range(args.gpus))
            try:
                if os.environ.get("BENCH_TEST_GROUP_FAILS") == "1":   # (tests/test_group.py: the hand-over below, without a broken node)
                    raise RuntimeError("BENCH_TEST_GROUP_FAILS=1")
                line = group_main(args, pkg, devices, shared, args.gpus)
            except Exception as ex:   # noqa: BLE001 - whatever it was, the other ranks must hear of it
                if world == 1:
                    raise
                group_error = f"{type(ex).__name__}: {ex}"
                sys.stderr.write(f"bench.py: the device group did not come up ({group_error}); falling back to one engine per rank\n")
        if world > 1:
            word = [group_error]
            dist.broadcast_object_list(word, src=0)
            group_error = word[0]
            dist.barrier()
            dist.destroy_process_group()
        if group_error is None:
            if rank == 0:
                try:
                    import ctypes
                    ctypes.CDLL(None).fflush(None)
                except Exception:
                    pass
                print(json.dumps(line), flush=True)
                if not line["verified"]:
                    raise SystemExit(f"bench.py: the timed kernel's output differs from the oracle: {line['verification']}")
            return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the multifm engine has no CPU path")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 runs the RCCL path (process group, in-place broadcast into the engine buffer) even
    # with one rank, so the N>1 plumbing can be exercised on a single-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # no version banner on stdout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    cpg = args.channels_per_gpu or 64
    total_ch = cpg * world
    fs, decim, taps, all_offs, all_gains = pkg.synth.plan(args.config, nr_channels=total_ch)
    lo, hi = pkg.dist.shard_range(total_ch, rank, world)
    offs, gains = all_offs[lo:hi], all_gains[lo:hi]
    block = 1 << args.block_log2
    T = len(taps)

    # engine input buffers are torch tensors so RCCL can write straight into them
    lib = pkg.load_library()
    in_bytes = lib.mfm_engine_input_bytes(block, T)
    bufs = [torch.empty(in_bytes // 2, dtype=torch.int16, device="cuda") for _ in range(2)]
    eng = pkg.Engine(fs, decim, block, device=local_rank,
                     # one launch in four carries the event pair; runs of fewer than 64 steps time every launch (the driver's 20)
                     flags=pkg.binding.MFM_F_DEVICE_ONLY | pkg.binding.MFM_F_TIMING |
                     (pkg.binding.MFM_F_TIMING_SPARSE if sparse_timing(args.steps) else 0) |
                     (pkg.binding.MFM_F_OVERLAP if args.overlap else 0) |
                     (pkg.binding.MFM_F_FORCE_DOT2 if args.kernel == "dot2" else 0) |
                     (pkg.binding.MFM_F_FORCE_MFMA_V1 if args.kernel == "mfma1" else 0) |
                     (pkg.binding.MFM_F_STREAM_TAPS if args.kernel == "mfma1s" else 0) |
                     (pkg.binding.MFM_F_V3L_ONE_ROW_BLOCK if args.kernel == "v3l1" else 0) |
                     (pkg.binding.MFM_F_SLICE_64 if args.kernel == "slice64" else 0) |
                     (pkg.binding.MFM_F_SLICE_128 if args.kernel == "slice128" else 0) |
                     (pkg.binding.MFM_F_PCM_WRITE_BACK if args.pcm_write_back else 0),
                     ext_input=(bufs[0].data_ptr(), bufs[1].data_ptr()))
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()

    # synthetic wideband IQ, resident in HBM before the timed region (rank 0 is the ingest GPU)
    in8 = args.input == "rtlsdr_u8"
    if rank == 0:
        base = pkg.synth.synth_iq(1 << 22, fs, all_offs[:: max(1, total_ch // 8)][:8], seed=7)
        if in8:
            # what the dongle would have delivered for this signal: unsigned bytes, 127 = zero
            u8 = np.clip((base.astype(np.int32) >> 7) + 127, 0, 255).astype(np.uint8)
            reps = -(-(in_bytes // 2) // u8.shape[0])
            host8 = np.tile(u8, (reps, 1))[: in_bytes // 2].reshape(-1)
            for b in bufs:
                b.view(torch.uint8)[: host8.size].copy_(torch.from_numpy(host8))
        else:
            reps = -(-(in_bytes // 4) // base.shape[0])
            host = np.tile(base, (reps, 1))[: in_bytes // 4].reshape(-1)
            for b in bufs:
                b.copy_(torch.from_numpy(host))
    torch.cuda.synchronize()

    # the exchange step: scatter + all-gather by default (tsl-sdr_amd/dist.py; DESIGN.md section 7: a broadcast delivers at
    # one xGMI link's rate per GPU, the all-gather uses all of them); MFM_EXCHANGE=broadcast pins the other form, =auto times
    # both once on the real buffer and keeps the faster
    exchange = pkg.dist.BlockExchange(src=0, algo=os.environ.get("MFM_EXCHANGE", "scatter_allgather")) if use_dist else None

    def step():
        ptr, cap = eng.acquire_input_bytes(pkg.binding.MFM_IN_RTLSDR_U8) if in8 else eng.acquire_input()
        which = 0 if ptr < bufs[0].data_ptr() + in_bytes and ptr >= bufs[0].data_ptr() else 1
        off = (ptr - bufs[which].data_ptr()) // 2
        if use_dist:
            # int16 elements of the view: 2 per sample, 1 per sample when the block travels as bytes
            view = bufs[which][off: off + (block if in8 else 2 * block)]
            if exchange.algo == "auto":
                exchange.choose(view, sync=torch.cuda.synchronize)
            exchange.run(view)
        # the RCCL broadcast runs on torch's stream; without it (N = 1) the block is already in place
        eng.submit(block, producer_stream=torch.cuda.current_stream().cuda_stream, wait_producer=use_dist)

    def fence():
        torch.cuda.synchronize()
        eng.sync()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # settle: the board's power management takes ~100 launches to reach the sustained clock (profiles/README.md);
    # every rank runs the same number of steps (rank 0 decides), so the exchange stays in lock-step
    settle_steps = 0
    if args.settle_seconds > 0:
        fence()
        t_s = time.perf_counter()
        for _ in range(8):
            step()
        fence()
        per = max((time.perf_counter() - t_s) / 8, 1e-6)
        settle_steps = int(min(20000, max(0, args.settle_seconds / per)))
        if use_dist:
            t = torch.tensor([settle_steps], device="cuda")
            dist.broadcast(t, src=0)
            settle_steps = int(t.item())
        for _ in range(settle_steps):
            step()
        fence()
    # a fresh stream from here (history, rotators, discriminator state): the self-check behind the timed region steps the
    # oracle's rotators to the last block's first output, and that distance should be the run's warm-up + timed launches,
    # not the settle phase's thousands.  Same kernel, same launch geometry, nothing inside the timed region changes.
    eng.reset()
    for _ in range(args.warmup):
        step()
    fence()
    st0 = eng.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    st1 = eng.stats()
    cycles = launch_clocks(eng, min(args.steps, 1024)) if rank == 0 else None
    if use_dist:
        dt = pkg.dist.max_over_ranks(dt, device="cuda")

    launches = st1["launches"] - st0["launches"]
    timed = st1["timed_launches"] - st0["timed_launches"]  # MFM_F_TIMING_SPARSE: one launch in four carries the event pair
    k_ms = (st1["kernel_ms"] - st0["kernel_ms"]) / max(1, timed)
    per_launch = np.sort(eng.launch_ms(min(int(timed), 4096)).astype(np.float64))
    outs = (st1["outputs"] - st0["outputs"]) / max(1, launches)
    bytes_per_launch = block * (2 if in8 else 4) + len(offs) * outs * 2  # 4 (2 as bytes) + 2*C_g/D bytes per input sample
    dot2_per_launch = 2.0 * len(offs) * T * outs               # two v_dot2 lane-ops per complex tap per output
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    mfma = st1["kernel_variant"] >= 1
    kname = {0: "mfm_channel_kernel", 1: "mfm_channel_kernel_mfma", 2: "mfm_channel_kernel_v3"}[st1["kernel_variant"]]
    if mfma:
        # exact int16 MACs done as four int8 byte-plane products on the matrix cores: 2 ops x 4 planes x (4 real MACs per
        # complex tap).  That is the four-plane figure; the kernel does not issue the two products of a k-step whose
        # high-byte tap plane is all zero (mfm_stats.tap_hi_mask), so the matrix pipe's real load is `frac_issued`.
        ops = 2.0 * 4.0 * 4.0 * len(offs) * T * outs
        ks, hi = max(1, st1["k_steps"]), bin(st1["tap_hi_mask"]).count("1")
        issued_share = (2.0 * ks + 2.0 * hi) / (4.0 * ks)
        compute_roof = {"bound": "mfma_i8", "achieved": ops / (k_ms * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS,
                        "unit": "TOP/s", "frac": ops / (k_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                        "mfma_per_k_step_issued": 2 + 2.0 * hi / ks, "mfma_per_k_step_four_planes": 4,
                        "frac_issued": ops * issued_share / (k_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS}
    else:
        compute_roof = {"bound": "v_dot2_i32_i16", "achieved": dot2_per_launch / (k_ms * 1e-3) / 1e12,
                        "peak": VALU_DOT2_PEAK / 1e12, "unit": "T lane-ops/s",
                        "frac": dot2_per_launch / (k_ms * 1e-3) / VALU_DOT2_PEAK}
    msamp = args.steps * block / dt / 1e6

    # HBM bytes per launch from the committed PMC passes (profiles/, tools/prof_r02.sh: FETCH_SIZE and WRITE_SIZE each in
    # its own rocprofv3 --pmc run, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): only quoted for the
    # exact workload and kernel they were collected on
    traffic, traffic_source = None, None
    import glob as _glob
    rounds = sorted({os.path.basename(f)[:3] for f in _glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json"))}, reverse=True)
    for rnd in rounds:   # the newest round's passes first
        tpath = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")
        if st1["kernel_variant"] == 2 and world == 1 and args.block_log2 == 26 and cpg == 64 and not in8 and \
                args.config == "cfg2_64ch" and os.path.exists(tpath):
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
            traffic_source = f"profiles/{rnd}_hbm_traffic.json (rocprofv3 --pmc passes of this command, not this run)"
            break

    # the issue ceiling of this formulation on this chip: matrix and other vector instructions of a SIMD do not overlap on gfx950,
    # so a launch cannot be shorter than their issue cycles; what the kernel takes beyond that is latency it does not hide.  The
    # HBM roof (0.70 in the north star) is not what bounds this kernel - this is.  Cycles and clock: this run's, stamped by the
    # kernel; instruction counts: of this library and kernel instance (issue_model()).
    ceiling = None
    if rank == 0 and st1["kernel_variant"] == 2 and world == 1:
        ceiling = issue_model(instance_name(pkg, st1, in8), st1, cycles, achieved / HBM_PEAK_GBPS, library_sha16(pkg))

    # (the self-check first: it reads what the LAST TIMED launch left in HBM)
    verified = None
    if rank == 0:
        n_last = eng.last_output_device()[2]
        verified = verify_last_block(pkg, eng, fs, decim, taps, offs, gains, eng.stats()["outputs"] - n_last)

    # One sysfs sample of the board's clock and power, taken while the same steps keep running right BEHIND the timed region: a
    # hwmon read is a message to the SMU that holds the submission up for ~1.5 ms - inside a 2.4 ms timed region it cost a
    # third of `value` (tools/r05/step_overheads.sh: ms_per_step 0.121 -> 0.180).  The clock the timed launches really ran
    # at is in `clocks` (stamped by the kernel itself); this is the board's own reading of the same sustained state.
    smi = None
    if os.environ.get("BENCH_NO_BOARD_SAMPLE") != "1":
        sampler = BoardSampler(local_rank) if rank == 0 else None
        # The read starts only after BOARD_SAMPLE_AFTER_S of back-to-back steps - the host work between the timed region and
        # this loop (the self-check: tenths of a second) lets the board's power reading, a moving average over a few hundred ms,
        # fall to a transitional value (854 W read 0.4 ms into the resumed load; 1375-1387 W sustained:
        # profiles/r05_power_trace.txt) - and the workload keeps running for as long as the read takes (2-4 ms: a reading taken
        # after the queue has drained shows the idle clock).  Step counts come from the timed region's own step time, which is
        # the same number on every rank (max over ranks): with N > 1 a step is a collective.
        step_s = max(dt / max(1, args.steps), 1e-6)
        n_pre = int(min(20000, max(16, BOARD_SAMPLE_AFTER_S / step_s)))
        n_post = int(min(4000, max(16, 0.03 / step_s)))
        for _ in range(n_pre):
            step()
        torch.cuda.synchronize()
        eng.sync()   # the device has now run n_pre steps back to back (the host submits faster than it executes) ...
        for _ in range(16):
            step()   # ... and is kept busy while the read starts
        if sampler:
            sampler.start()
        for _ in range(n_post):
            step()
        fence()
        if sampler:
            smi = sampler.result()
            if smi is not None:
                smi["when"] = "after %.1f s of further back-to-back steps of the same workload behind the timed region" % BOARD_SAMPLE_AFTER_S

    if rank == 0:
        line = {
            "metric": "input IQ MSamp/s x channels demodulated",
            "value": msamp * total_ch,
            "unit": "MSamp/s x channels",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16" if not in8 else "int16 (8-bit input read as bytes)", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cpg} FM channels per GPU ({total_ch} total), {T}-tap 25 kHz LPF, "
                                   f"decimation {decim}, fs {fs} Hz-shaped int16 IQ, block 2^{args.block_log2} samples",
                       "channels_per_gpu": cpg, "channels_total": total_ch, "block_samples": block,
                       "decimation": int(decim), "taps": int(T), "sample_rate_hz": int(fs),
                       "input_msamp_per_s": msamp,
                       "parallelism": "1 GPU" if not use_dist else
                                      f"channel shards x{world} + RCCL {exchange.algo} of the IQ block"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "timed_launches": int(timed), "launches": int(launches),
                         "kernel": kname, "slice_channels": st1.get("slice_channels"), "kernel_ms": k_ms,
                         "kernel_ms_min": float(per_launch[0]) if len(per_launch) else None,
                         "kernel_ms_median": float(np.median(per_launch)) if len(per_launch) else None,
                         "kernel_ms_p95": float(np.percentile(per_launch, 95)) if len(per_launch) >= 20 else None,
                         "bytes_per_launch": bytes_per_launch,
                         # what holds the kernel below the HBM roof (SQ counters in profiles/)
                         "binding": "simd_issue" if mfma else "valu_dot2",
                         "ceiling_frac": ceiling["ceiling_frac"] if ceiling else None, "issue_model": ceiling,
                         "instance": instance_name(pkg, st1, in8), "library_sha16": library_sha16(pkg),
                         # the kernel's own clocks for the timed launches, and one sysfs sample taken while they ran
                         "clocks": cycles, "board_sample": smi,
                         "energy": energy_figures(smi, dt / args.steps * 1e3, len(offs), outs),
                         # the same three as plain numbers (a record that keeps only scalars of this object still says which regime
                         # the run was in: the clock inside the timed launches, the board's power against its cap, SIMDs busy)
                         "sclk_mhz_in_launches": (cycles or {}).get("sclk_mhz_effective"),
                         "board_power_w": (smi or {}).get("power_w"), "board_power_cap_w": (smi or {}).get("power_cap_w"),
                         "simd_busy_fraction": (ceiling or {}).get("simd_busy_fraction")},
            "verified": verified["verified"], "verification": verified,
            "rotators": {"exact_channels": st1["rot_exact_channels"], "channels": len(offs)},
            "protocol": {"settle_seconds": args.settle_seconds, "settle_steps": settle_steps + 8,
                         "warmup_steps": args.warmup, "timed_steps": args.steps},
            "compute_roofline": compute_roof,
            # N > 1: every step moves the whole block to each of the other N - 1 GPUs; the bench replays blocks as fast
            # as the GPUs take them, so this - not the kernel - is what an N > 1 line is usually bound by (a live
            # 2.4 MS/s stream is 10 MB/s).  xGMI: 7 links x ~153 GB/s per GPU, point to point.
            # set when `--exchange group` (the default at N > 1) could not bring the device group up on this node and the ranks went
            # on with one engine each: what the group said
            "group_error": group_error,
            "exchange": None if not use_dist else {
                "algo": exchange.algo, "algo_timings_s": exchange.timings, "bytes_per_step_per_peer": block * (2 if in8 else 4),
                "peers": world - 1,
                # what every peer would have to receive for the exchange to hide behind the kernel (a step then costs
                # max(exchange, kernel)): one block per kernel time
                "needed_GBps_per_peer": block * (2 if in8 else 4) / (k_ms * 1e-3) / 1e9,
                "delivered_GBps_per_peer": block * (2 if in8 else 4) / (dt / args.steps) / 1e9,
                # does the exchange hide behind the kernel (a step costs max(exchange, kernel))?
                "hidden": bool(dt / args.steps <= 1.1 * k_ms * 1e-3),
                "delivered_GBps_total": block * (2 if in8 else 4) * (world - 1) / (dt / args.steps) / 1e9,
                "xgmi_link_peak_GBps": 153.0, "xgmi_links_per_gpu": 7},
            "geometry": {"outputs_per_tile": st1["outputs_per_tile"], "lds_bytes": st1["lds_bytes"],
                         "grid": st1["grid_last"], "rot_table_entries": st1["rot_table_entries"],
                         "k_steps": st1["k_steps"], "tap_hi_mask": st1["tap_hi_mask"], "taps_resident": st1["taps_resident"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, fs, decim, taps, offs, gains, args.cpu_seconds)

    eng.close()
    if rank == 0 and world == 1 and not args.no_fp32:
        # BASELINE configs[4] "fp32 vs int16 IQ path": the same channels on float32 IQ, outside the timed region,
        # reported next to the headline (never part of `value`)
        line["fp32_iq_path"] = fp32_path(pkg, torch, fs, decim, taps, offs, gains, line["roofline"]["kernel_ms"],
                                         block, block_log2=args.block_log2)
    if rank == 0 and world == 1 and not args.no_series:
        # SURVEY.md 8(d)'s own protocol (2^20-sample blocks, >= 64 per timing; host-fed end to end), outside the timed region
        line["block_series"] = block_series(pkg, torch, fs, decim, taps, offs, gains)
        line["end_to_end"] = end_to_end(pkg, fs, decim, taps, offs, gains)
    if rank == 0 and world == 1 and not args.no_chain and decim == 96:
        line["flex_chain"] = flex_chain(pkg, torch, fs, decim, taps, offs, gains, block)
        if line["roofline"]["kernel"].startswith("mfm_channel_kernel_v3"):
            line["ingest_8bit"] = ingest_8bit(pkg, fs, decim, taps, offs, gains, block, line["roofline"]["kernel_ms"])
        line["other_geometries"] = other_geometries(pkg, torch, block)
        line["north_star_shape"] = north_star_shape(pkg, torch, block)
        if not in8 and args.kernel == "auto" and not args.overlap:
            # the same workload through the product's device-group path (one device: no exchange): must agree with `value`
            try:
                gr = group_run(pkg, args, [local_rank], False, fs, decim, taps, offs, gains, block, args.steps, args.warmup, 0.25)
                gv = args.steps * block / gr["dt"] / 1e6 * total_ch
                line["group_path"] = {"value": gv, "unit": "MSamp/s x channels", "ms_per_step": gr["dt"] / args.steps * 1e3,
                                      "ratio_to_value": gv / line["value"], "kernel_ms": gr["shards"][0]["kernel_ms"],
                                      "verified": gr["verified"]["verified"], "shards": gr["shards"], "exchange_info": gr["exchange"],
                                      "how": "mfm_group_acquire_input + mfm_group_submit on one device, same blocks, same step counts"}
            except Exception as e:
                line["group_path"] = {"error": repr(e)}
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would come out
        # after anything Python prints: flush it first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
        if not line["verified"]:
            raise SystemExit(f"bench.py: the timed kernel's output differs from the oracle: {line['verification']}")


if __name__ == "__main__":
    main()
