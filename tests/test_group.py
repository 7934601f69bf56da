"""The device group (mfm_group_*): one channel set on several GPUs behind the C ABI (SURVEY.md section 8b "set_devices",
section 8e).  Shard arithmetic and argument checks on the CPU; on the GPU box (one device) the group is run through both
of its ingest paths - direct staging, and the RCCL path a multi-GPU group uses (ncclCommInitAll, in-place ncclBroadcast
into the engines' input buffers, submit ordered behind it) - and compared with the oracle."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_tile_the_channel_set_like_the_python_side(pkg):
    for n in (1, 2, 63, 64, 65, 1000, 1024, 2048):
        for g in (1, 2, 3, 4, 8, 16):
            got = []
            for s in range(g):
                lo, cnt = pkg.binding.shard_range(n, g, s)
                assert n // g <= cnt <= -(-n // g)
                got.extend(range(lo, lo + cnt))
                # the C host and bench.py (torch.distributed ranks) must agree on who owns which channel
                assert (lo, lo + cnt) == tuple(pkg.dist.shard_range(n, s, g))
            assert got == list(range(n))


def test_group_argument_checks(pkg):
    b = pkg.binding
    lib = pkg.load_library()
    for devices, ok in (((), False), ((0, 0), False), (tuple(range(17)), False)):
        cfg = b.GroupConfig()
        cfg.abi_version = b.MFM_ABI_VERSION
        cfg.nr_devices = len(devices)
        for i, d in enumerate(devices[:16]):
            cfg.devices[i] = d
        cfg.sample_rate_hz, cfg.decimation, cfg.max_block_samples = 1000000, 40, 4096
        h = C.c_void_p()
        assert (lib.mfm_group_create(C.byref(h), C.byref(cfg)) == 0) == ok
    b.Group(1000000, 40, 4096, devices=(0,), flags=b.MFM_F_DEVICE_ONLY).close()   # resident input: valid since round 5
    g = b.Group(1000000, 40, 4096, devices=(0, 1))
    with pytest.raises(b.MfmError):
        g.add_channel(1000, np.ones(8))          # taps < decimation: rejected like the engine does
    with pytest.raises(b.MfmError):
        g.commit()                               # no channels
    g.close()


def _drive(grp, iq, block, pkg):
    parts, pos = [], 0
    while pos < len(iq):
        m = min(block, len(iq) - pos)
        rc = grp.push(iq[pos:pos + m])
        if rc == pkg.binding.MFM_E_BUSY:
            parts.append(grp.fetch()[1])
            continue
        pos += m
    grp.sync()
    while True:
        got = grp.fetch()
        if got is None:
            break
        parts.append(got[1])
    return np.concatenate(parts, axis=1)


def _xmode(b, exchange):
    return {"direct": b.MFM_X_AUTO, "rccl": b.MFM_X_RCCL, "allgather": b.MFM_X_RCCL_ALLGATHER}[exchange]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["direct", "rccl", "allgather"])
def test_group_on_one_device_matches_oracle(pkg, ora, exchange):
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=70)
    iq = pkg.synth.synth_iq(96 * 3000 + 321, fs, offs[:4], seed=7)
    b = pkg.binding
    grp = b.Group(fs, decim, 1 << 16, devices=(0,), exchange=_xmode(b, exchange))
    for o, g in zip(offs, gains):
        grp.add_channel(int(o), taps, float(g))
    grp.commit()
    assert grp.nr_shards == 1 and grp.shard_info(0) == (0, 70, 0)
    pcm = _drive(grp, iq, 50000, pkg)
    uses, blocks, moved = grp.exchange_info()
    grp.close()
    assert uses == (exchange != "direct") and (blocks > 0) == uses and moved == 0  # one device: nothing leaves it
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    assert pcm.shape == ref.shape and np.array_equal(pcm, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["direct", "rccl", "allgather"])
def test_group_exchanges_8bit_blocks_as_bytes(pkg, ora, exchange):
    """RTL-SDR bytes through the group: where every member's kernel can read them as they are the block is staged,
    broadcast (RCCL path: half the bytes of an int16 block) and consumed as bytes; a stream that turns to int16 goes
    on through the widening.  PCM against the oracle on the reference's host-side widening."""
    fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=24)
    rng = np.random.RandomState(9)
    b = pkg.binding
    grp = b.Group(fs, decim, 1 << 16, devices=(0,), exchange=_xmode(b, exchange))
    for o, g in zip(offs, gains):
        grp.add_channel(int(o), taps, float(g))
    grp.commit()
    blocks = [(rng.randint(0, 256, size=(m, 2)).astype(np.uint8), 3) for m in (65536, 30001, 4096, 50000)]
    blocks.append((rng.randint(-32768, 32768, size=(20000, 2)).astype(np.int16), 0))
    blocks.append((rng.randint(0, 256, size=(40000, 2)).astype(np.uint8), 3))
    iq, parts = [], []
    for blk, fmt in blocks:
        iq.append(blk.reshape(-1, 2) if fmt == 0 else ora.unpack_bytes(blk, fmt).reshape(-1, 2))
        while grp.push(blk, fmt) == b.MFM_E_BUSY:
            parts.append(grp.fetch()[1])
    grp.sync()
    while True:
        got = grp.fetch()
        if got is None:
            break
        parts.append(got[1])
    st = grp.stats(0)
    _, nblk, moved = grp.exchange_info()
    grp.close()
    assert st["launches_8bit"] == 4 and st["launches"] == 6
    assert moved == 0 and (nblk == 6) == (exchange != "direct")
    iq = np.concatenate(iq)
    cre = np.stack([ora.make_taps(taps, int(o), fs, float(g))[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, int(o), fs, float(g))[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim, threads=8)
    pcm = np.concatenate(parts, axis=1)
    assert pcm.shape == ref.shape and np.array_equal(pcm, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shards,mode,nch", [(2, "rccl", 70), (3, "allgather", 70), (8, "allgather", 130), (8, "rccl", 64),
                                             (4, "allgather", 3), (2, "auto", 16)])
def test_group_of_several_shards_on_one_device_through_a_fake_transport(tmp_path, shards, mode, nch):
    """No multi-GPU node is available to this project, so the S > 1 paths of the device group are checked for CORRECTNESS
    on the one GPU of the box: MFM_F_GROUP_SHARED_DEVICE lets a group list the device several times, and a test double of
    the RCCL calls (tests/hoststub/fake_rccl.cpp, built as librccl.so into a directory put first in LD_LIBRARY_PATH of a
    fresh process) moves the bytes with device-to-device copies.  What that exercises for real: contiguous shard ranges,
    one engine per shard, the exchange's pointer arithmetic in both modes (which part of the block goes to which shard, the
    in-place all-gather, the remainder broadcast), 8-bit blocks travelling as bytes, submit order, per-shard fetch and the
    concatenated PCM - all against the oracle.  It says nothing about links or speed."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(["python3", os.path.join(ROOT, "tests", "hoststub", "multi_shard_run.py"), str(shards), mode, str(nch)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "multi-shard ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
def test_two_shard_group_pushed_and_fetched_from_two_threads(tmp_path):
    """a drain thread calling mfm_group_fetch between shard 0's and shard 1's submit got "shards out of
    step" and took the receiver down.  Real engines, two shards on the one device over the fake transport, 400 blocks of 2048
    samples pushed by one thread while another one fetches: only "nothing yet" or whole blocks, and the oracle's PCM."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(["python3", os.path.join(ROOT, "tests", "hoststub", "multi_shard_threads.py")], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0 and "threads ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "allgather"])
def test_multifm_driver_with_three_shards_on_one_device(tmp_path, pkg, ora, exchange):
    """multifm_amd with "gpuDevices": [0, 0, 0] (test aid "gpuTestSharedDevice") over the fake transport: the C host's
    multi-shard path - one submit thread pushing to the group, one drain thread fetching a block per shard and writing
    channel c from row c - first(shard) of its shard's block - with seven channels over three shards; every FIFO byte
    stream must be the oracle's PCM."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    fs, decim, center = 1000000, 40, 929500000
    offs = [112500, -200000, 3125, 250000, -25000, 0, 77777]
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_1000k_128.json")
    taps = np.array(json.load(open(taps_file))["lpfTaps"])
    n = 4096 * 33 + 1001
    iq = pkg.synth.synth_iq(n, fs, offs[:4], seed=15)
    cap = tmp_path / "cap.bin"
    cap.write_bytes(iq.tobytes())
    cfg = {"device": {"type": "file", "filename": str(cap), "fileFormat": "cs16"}, "sampleRateHz": fs, "centerFreqHz": center,
           "nrSampBufs": 16, "decimationFactor": decim, "gpuDevices": [0, 0, 0], "gpuTestSharedDevice": True,
           "gpuExchange": exchange, "channels": []}
    outs = []
    for i, f in enumerate(offs):
        o = tmp_path / f"ch{i}.pcm"
        o.write_bytes(b"")
        cfg["channels"].append({"outFifo": str(o), "chanCenterFreq": int(center + f)})
        outs.append(o)
    cj = tmp_path / "cfg.json"
    cj.write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "tsl-sdr_amd", "host", "multifm_amd")
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(cj), taps_file], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "sharded over 3 GPU(s)" in r.stderr
    cre = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim)
    for i, o in enumerate(outs):
        got = np.frombuffer(o.read_bytes(), dtype=np.int16)
        assert np.array_equal(got, ref[i]), i


@pytest.mark.gpu
def test_group_with_more_devices_than_the_box_has_fails_cleanly(pkg):
    import torch
    n = torch.cuda.device_count()
    g = pkg.binding.Group(1000000, 40, 4096, devices=tuple(range(n + 1)))
    for k in range(n + 1):
        g.add_channel(1000 * k, np.ones(64) / 64)
    with pytest.raises(pkg.binding.MfmError) as ei:
        g.commit()
    assert ei.value.code == pkg.binding.MFM_E_DEVICE
    g.close()


@pytest.mark.gpu
def test_multifm_driver_through_the_rccl_exchange(tmp_path, pkg, ora):
    """multifm_amd with "gpuDevices": [0] and "gpuExchange": "rccl": the C host's multi-device ingest path (submit
    thread -> mfm_group_push -> stage on the root, ncclBroadcast, submit) on the one GPU of the box; FIFO byte streams
    must be the oracle's PCM."""
    fs, decim, center = 1000000, 40, 929500000
    offs = [112500, -200000, 3125]
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_1000k_128.json")
    taps = np.array(json.load(open(taps_file))["lpfTaps"])
    n = 4096 * 21 + 77
    iq = pkg.synth.synth_iq(n, fs, offs, seed=5)
    cap = tmp_path / "cap.bin"
    cap.write_bytes(iq.tobytes())
    cfg = {"device": {"type": "file", "filename": str(cap), "fileFormat": "cs16"}, "sampleRateHz": fs, "centerFreqHz": center,
           "nrSampBufs": 16, "decimationFactor": decim, "gpuDevices": [0], "gpuExchange": "rccl", "channels": []}
    outs = []
    for i, f in enumerate(offs):
        o = tmp_path / f"ch{i}.pcm"
        o.write_bytes(b"")
        cfg["channels"].append({"outFifo": str(o), "chanCenterFreq": int(center + f)})
        outs.append(o)
    cj = tmp_path / "cfg.json"
    cj.write_text(json.dumps(cfg))
    exe = os.path.join(ROOT, "tsl-sdr_amd", "host", "multifm_amd")
    r = subprocess.run([exe, str(cj), taps_file], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RCCL exchange forced" in r.stderr
    cre = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[0] for o in offs])
    cim = np.stack([ora.make_taps(taps, int(o), fs, 1.0)[1] for o in offs])
    incr = np.stack([ora.rot_incr(int(o), fs, decim) for o in offs])
    ref, _ = ora.run_channels(iq, cre, cim, incr, decim)
    for i, o in enumerate(outs):
        got = np.frombuffer(o.read_bytes(), dtype=np.int16)
        assert np.array_equal(got, ref[i]), i


@pytest.mark.gpu
def test_bench_runs_its_distributed_path_on_one_rank(pkg):
    """bench.py with BENCH_FORCE_DIST=1: process group on RCCL, in-place broadcast of the block into the engine's
    input buffer, submit behind it - the N > 1 code path of the bench on the one GPU of the box."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--settle-seconds", "0", "--no-cpu-baseline", "--no-fp32", "--block-log2", "22"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL" in line["config"]["parallelism"]
    assert line["verified"] is True and line["exchange"]["needed_GBps_per_peer"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 4])
def test_bench_group_path_over_the_fake_transport(tmp_path, shards):
    """bench.py --gpus N goes through the PRODUCT's multi-GPU code (mfm_group_acquire_input / mfm_group_submit: channel shards, the
    library's RCCL scatter + all-gather of every block, one process).  No multi-GPU node here: the same call path with
    --group-shards S puts S shards on the one device (MFM_F_GROUP_SHARED_DEVICE) over the test double of the RCCL calls.  The
    line must carry the exchange's counters and every shard's statistics, the shards must have moved in lock step, and
    the self-check (EVERY shard's last launch against the oracle, on the input that shard's launch read) must hold."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--group-shards", str(shards), "--channels-per-gpu", "40",
                        "--steps", "5", "--warmup", "2", "--settle-seconds", "0", "--block-log2", "21"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    g = line["group"]
    assert line["verified"] is True and line["value"] > 0 and "mfm_group" in line["config"]["parallelism"]
    assert len(g["shards"]) == shards and sum(sh["channels"] for sh in g["shards"]) == 40 * shards
    assert g["exchange_info"]["uses_rccl"] and g["exchange_info"]["bytes_to_other_devices"] >= 5 * (shards - 1) * (4 << 21)
    assert len({sh["launches"] for sh in g["shards"]}) == 1 and all(sh["launches"] == 5 and sh["kernel_variant"] == 2 for sh in g["shards"])
    assert len({sh["pending_blocks"] for sh in g["shards"]}) == 1  # (device-only: nothing is fetched; the shards stay in step)
    # what makes an N > 1 line readable: which RCCL, how many ranks it reports, every shard's device,
    # exchange time against kernel time, and the same per-GPU shape on one device of the same run
    x = g["exchange_info"]
    assert x["rccl_ranks"] == shards and os.path.realpath(x["rccl_library"]) == os.path.realpath(so), x
    assert len(x["per_shard"]) == shards
    for d in x["per_shard"]:
        assert d["rccl_ranks"] == shards and d["pci"] and d["timed_exchanges"] >= 1 and d["exchange_ms"] > 0.0, d
        assert d["kernel_ms"] and d["kernel_ms"] > 0.0 and d["bound"] in ("kernel", "exchange"), d
    assert line["verification"]["shards_verified"] == [True] * shards   # EVERY shard's last launch against the oracle
    one = line["single_gpu_same_shape"]
    assert one["channels"] == 40 and one["value"] > 0 and one["verified"] is True, one
    assert abs(line["scaling_efficiency"] - line["value"] / (shards * one["value"])) < 1e-9 and line["config"]["channels_per_gpu"] == 40


@pytest.mark.gpu
def test_bench_group_path_as_the_driver_launches_it(tmp_path):
    """the driver's launch form for N > 1 - `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` - with two ranks on
    the one GPU of the box: rank 0 drives both shards (one device, fake transport), rank 1 waits for its word over gloo without
    touching the GPU; ONE JSON line, exit code 0."""
    so = tmp_path / "librccl.so"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", str(so),
                        os.path.join(ROOT, "tests", "hoststub", "fake_rccl.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    # (rank 0 has torch.distributed - and with it PyTorch's librccl - mapped by the time the group loads RCCL: the double is named)
    env = dict(os.environ, MFM_RCCL_LIBRARY=str(so))
    r = subprocess.run(["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--group-shards", "2",
                        "--channels-per-gpu", "40", "--steps", "5", "--warmup", "2", "--settle-seconds", "0", "--block-log2", "21"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and len(line["group"]["shards"]) == 2
    assert line["verification"]["shards_verified"] == [True, True] and line["scaling_efficiency"] > 0


def test_bench_ranks_leave_the_group_path_together_when_it_fails():
    """N > 1 under torch.distributed.run, the device group does not come up (BENCH_TEST_GROUP_FAILS=1 stands for a node whose RCCL
    cannot make the communicator in one process): rank 0 tells the other ranks over gloo and ALL of them go on to the per-rank form.
    Here (no GPU in this container, or one on the box) that form stops at its own first check - on both ranks, with the reason of
    the fall-back in rank 0's stderr, and without anybody waiting for a barrier the other side never reaches."""
    env = dict(os.environ, BENCH_TEST_GROUP_FAILS="1", BENCH_WATCHDOG_S="240")
    r = subprocess.run(["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--settle-seconds", "0", "--block-log2", "20", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode != 0 and "giving up" not in out, out[-3000:]
    assert "the device group did not come up (RuntimeError: BENCH_TEST_GROUP_FAILS=1); falling back to one engine per rank" in out, out[-3000:]
    # both ranks reached the per-rank form: without a second GPU it ends at the device check / at set_device of rank 1
    assert out.count("bench.py needs a GPU") == 2 or "invalid device ordinal" in out or "device" in out.lower(), out[-3000:]


@pytest.mark.gpu
def test_bench_line_carries_the_group_path_and_the_north_star_shape():
    """the N = 1 line: `group_path` - the same workload through mfm_group_* on one device - the same rate as `value` (loosely here: small
    blocks; 2 % at the driver's size), and `north_star_shape` - 1024 channels on the one GPU - with its matrix-instruction bound"""
    r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--settle-seconds", "0.3",
                        "--no-cpu-baseline", "--no-fp32", "--no-series", "--block-log2", "24"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    gp, ns = line["group_path"], line["north_star_shape"]
    # (no bound on the ratio here: beside other tests on the same GPU either of the two timed regions can be slowed down several
    # times over; at the driver's sizes, alone on the box, it is 0.98 .. 1.01: profiles/r05_bench_lines.jsonl)
    assert gp["verified"] is True and gp["ratio_to_value"] > 0.0 and gp["shards"][0]["launches"] == 20, gp
    assert ns["channels"] == 1024 and ns["kernel_variant"] == 2 and 0.0 < ns["roofline"]["frac"] < ns["bound_frac"]["at_nominal_5000_tops"] < 1.0, ns
    clk = line["roofline"]["clocks"]
    assert clk and clk["launches"] == 20 and 500.0 < clk["sclk_mhz_effective"] < 3000.0, clk
    # one launch in four carries the HIP event pair (on every launch the packets cost 2 % of `value`); every launch is timed by the
    # kernel's own 100 MHz stamps
    assert line["roofline"]["timed_launches"] == 5 and clk["kernel_ms_by_stamps"]["launches"] == 20
    assert 0.0 < clk["kernel_ms_by_stamps"]["min"] <= clk["kernel_ms_by_stamps"]["median"] <= clk["kernel_ms_by_stamps"]["max"] < 10.0


@pytest.mark.gpu
def test_bench_exchanges_8bit_blocks_as_bytes(pkg):
    """bench.py --input rtlsdr_u8 on its distributed path (one rank): the block lies in the engine's buffer as the bytes an
    RTL-SDR delivers, the exchange moves 2 bytes per sample instead of 4, the matrix kernel reads the bytes, and the line's
    self-check widens the launch's input as multifm/rtl_sdr_if.c:146-148 does before it asks the oracle."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--settle-seconds", "0", "--no-cpu-baseline", "--no-fp32", "--no-chain", "--no-series", "--block-log2", "22",
                        "--input", "rtlsdr_u8"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["verified"] is True and line["exchange"]["bytes_per_step_per_peer"] == 2 << 22
    assert line["roofline"]["bytes_per_launch"] < 4 << 22
