"""The C host around the engine (tsl-sdr_amd/host): JSON configuration with the reference's key names,
the sample_buf pool and refcount contract, and - on the GPU - the multifm-shaped driver end to end on
file_if input (BASELINE configs[0])."""
import ctypes as C
import json
import re
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "tsl-sdr_amd", "host", "libmfm_host.so")
MULTIFM = os.path.join(ROOT, "tsl-sdr_amd", "host", "multifm_amd")
REF_ETC = "/root/reference/etc"


class Config(C.Structure):
    _fields_ = [("node", C.c_void_p), ("owner", C.c_bool)]


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(HOST_SO):
        pytest.fail(f"{HOST_SO} missing: run make -C tsl-sdr_amd")
    h = C.CDLL(HOST_SO)
    h.config_new.argtypes = [C.POINTER(C.POINTER(Config))]
    h.config_add.argtypes = [C.POINTER(Config), C.c_char_p]
    h.config_add_string.argtypes = [C.POINTER(Config), C.c_char_p]
    h.config_delete.argtypes = [C.POINTER(C.POINTER(Config))]
    h.config_delete.restype = None
    h.config_get.argtypes = [C.POINTER(Config), C.POINTER(Config), C.c_char_p]
    h.config_get_integer.argtypes = [C.POINTER(Config), C.POINTER(C.c_int), C.c_char_p]
    h.config_get_float.argtypes = [C.POINTER(Config), C.POINTER(C.c_double), C.c_char_p]
    h.config_get_string.argtypes = [C.POINTER(Config), C.POINTER(C.c_char_p), C.c_char_p]
    h.config_get_float_array.argtypes = [C.POINTER(Config), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t), C.c_char_p]
    h.config_array_length.argtypes = [C.POINTER(Config), C.POINTER(C.c_size_t)]
    h.config_array_at.argtypes = [C.POINTER(Config), C.POINTER(Config), C.c_size_t]
    return h


def _new(host):
    p = C.POINTER(Config)()
    assert host.config_new(C.byref(p)) == 0
    return p


def _int(host, cfg, key):
    v = C.c_int()
    rc = host.config_get_integer(cfg, C.byref(v), key.encode())
    return rc, v.value


def test_config_merge_and_types(host):
    cfg = _new(host)
    assert host.config_add_string(cfg, b'{"sampleRateHz": 1000000, "decimationFactor": 40, "x": {"y": [1, 2.5, -3e2]}}') == 0
    # a second "file" adds and overrides top-level keys (multifm.c:105-111 stacks config files)
    assert host.config_add_string(cfg, b'{"decimationFactor": 96, "lpfTaps": [0.25, 0.5, 0.25], "name": "a\\"b"}') == 0
    assert _int(host, cfg, "sampleRateHz") == (0, 1000000)
    assert _int(host, cfg, "decimationFactor") == (0, 96)
    assert _int(host, cfg, "missing")[0] < 0
    vals, n = C.POINTER(C.c_double)(), C.c_size_t()
    assert host.config_get_float_array(cfg, C.byref(vals), C.byref(n), b"lpfTaps") == 0
    assert [vals[i] for i in range(n.value)] == [0.25, 0.5, 0.25]
    s = C.c_char_p()
    assert host.config_get_string(cfg, C.byref(s), b"name") == 0 and s.value == b'a"b'
    sub, arr, item = Config(), Config(), Config()
    assert host.config_get(cfg, C.byref(sub), b"x") == 0
    assert host.config_get(C.byref(sub), C.byref(arr), b"y") == 0
    ln = C.c_size_t()
    assert host.config_array_length(C.byref(arr), C.byref(ln)) == 0 and ln.value == 3
    assert host.config_array_at(C.byref(arr), C.byref(item), 3) < 0
    # a float is not an integer (the reference's config_get_integer rejects 2.5 as well)
    assert host.config_add_string(cfg, b'{"f": 2.5}') == 0 and _int(host, cfg, "f")[0] < 0
    assert host.config_add_string(cfg, b'{"broken": [1, 2') < 0
    host.config_delete(C.byref(cfg))


def test_own_etc_files_load(host):
    cfg = _new(host)
    assert host.config_add(cfg, os.path.join(ROOT, "etc", "multifm_1ch_file.json").encode()) == 0
    assert host.config_add(cfg, os.path.join(ROOT, "etc", "lpf_25khz_1000k_128.json").encode()) == 0
    vals, n = C.POINTER(C.c_double)(), C.c_size_t()
    assert host.config_get_float_array(cfg, C.byref(vals), C.byref(n), b"lpfTaps") == 0 and n.value == 128
    assert abs(sum(vals[i] for i in range(128)) - 1.0) < 1e-12
    host.config_delete(C.byref(cfg))


@pytest.mark.skipif(not os.path.isdir(REF_ETC), reason="reference tree not present (only in the authoring container)")
def test_reference_etc_files_load_unchanged(host):
    """Every non-empty JSON the reference ships must load with its keys readable."""
    seen = 0
    for name in sorted(os.listdir(REF_ETC)):
        path = os.path.join(REF_ETC, name)
        if os.path.getsize(path) == 0:
            continue  # etc/pocsag_narrow.json and etc/pocsag_1200khz_fs.json are empty files
        want = json.load(open(path))
        cfg = _new(host)
        assert host.config_add(cfg, path.encode()) == 0, name
        for key, val in want.items():
            if isinstance(val, bool):
                continue
            if isinstance(val, int):
                assert _int(host, cfg, key) == (0, val), (name, key)
            elif isinstance(val, list) and val and isinstance(val[0], float):
                vals, n = C.POINTER(C.c_double)(), C.c_size_t()
                assert host.config_get_float_array(cfg, C.byref(vals), C.byref(n), key.encode()) == 0
                assert [vals[i] for i in range(n.value)] == val, (name, key)
            elif key == "channels":
                arr, item = Config(), Config()
                assert host.config_get(cfg, C.byref(arr), b"channels") == 0
                for i, ch in enumerate(val):
                    assert host.config_array_at(C.byref(arr), C.byref(item), i) == 0
                    assert _int(host, C.byref(item), "chanCenterFreq") == (0, ch["chanCenterFreq"])
                    s = C.c_char_p()
                    assert host.config_get_string(C.byref(item), C.byref(s), b"outFifo") == 0
                    assert s.value.decode() == ch["outFifo"]
        host.config_delete(C.byref(cfg))
        seen += 1
    assert seen >= 8


def test_deliver_never_blocks_and_a_stalled_device_drops_at_the_pool(tmp_path):
    """receiver_sample_buf_deliver() "must never block the front-end thread longer than a
    memcpy".  The receiver is driven on a test double of the device group whose push() reports MFM_E_BUSY (a stalled
    device / drain): deliver() keeps returning at once, the submit thread holds the queued buffers, the pool (8 frames)
    runs dry and receiver_sample_buf_alloc() drops + counts like multifm/receiver.c:57-63; when the device side comes
    back everything queued goes through and the pool refills."""
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    exe = tmp_path / "stall_test"
    srcs = [os.path.join(host, f) for f in ("mfm_tsl.c", "mfm_config.c", "mfm_receiver.c")]
    srcs += [os.path.join(ROOT, "tests", "hoststub", f) for f in ("stub_group.c", "stall_main.c")]
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-D_GNU_SOURCE", "-I" + host, "-I" + os.path.join(ROOT, "include"),
                        "-o", str(exe)] + srcs + ["-lpthread", "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sink = tmp_path / "pcm.out"
    sink.write_bytes(b"")
    r = subprocess.run([str(exe), str(sink)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    # 8 frames: the submit thread holds one (retrying its push), seven more wait in the ring; the other 32 reads are dropped
    assert res["delivered"] == 8 + 1 and res["dropped"] == 32 and res["alloc_fails"] == 32
    assert res["pushed_while_stalled"] == 0 and res["busy_returns"] > 0
    assert res["order_errors"] == 0 and res["pushed"] == res["delivered"], res  # runs: every buffer once, in order, from its own address
    assert res["pushed"] == 9 and res["samples"] == 9 * 4096 and res["pool_back"] == 1
    # deliver() queues a pointer: microseconds even on a loaded CI host, never the 200-us retry period of the old loop
    assert res["worst_deliver_ns_stalled"] < 100_000, res


def test_receiver_threads_are_race_free_under_tsan(tmp_path):
    """the same scenario (front end's thread, submit thread, drain thread, a device double that stalls and recovers) built
    with -fsanitize=thread: the ring, the failure flag, the shutdown flags and the counters are C11 atomics, buffers change
    hands through the ring / the pool's mutex - ThreadSanitizer must have nothing to report"""
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    exe = tmp_path / "stall_tsan"
    srcs = [os.path.join(host, f) for f in ("mfm_tsl.c", "mfm_config.c", "mfm_receiver.c")]
    srcs += [os.path.join(ROOT, "tests", "hoststub", f) for f in ("stub_group.c", "stall_main.c")]
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-D_GNU_SOURCE", "-I" + host,
                        "-I" + os.path.join(ROOT, "include"), "-o", str(exe)] + srcs + ["-lpthread", "-lm"],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    sink = tmp_path / "pcm.out"
    sink.write_bytes(b"")
    # (tools/sanitize_cpu.sh runs this suite with libasan preloaded; two sanitizer runtimes do not share a process)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    r = subprocess.run([str(exe), str(sink)], capture_output=True, text=True, timeout=120,
                       env=dict(env, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["delivered"] == 9 and res["dropped"] == 32 and res["pushed"] == 9 and res["pool_back"] == 1


@pytest.mark.parametrize("ending", ["reader_returns", "sigint", "file_eof"])
def test_driver_and_front_end_threads_are_race_free_under_tsan(tmp_path, ending):
    """The whole driver (host/multifm_main.c: configuration, front-end table, signal handling, cleanup order) with the RTL-SDR
    front end (host/mfm_rtl_sdr_if.c) on the device double, built with -fsanitize=thread: librtlsdr's reader thread (the test
    double replays a capture through the async callback) delivers into the receiver's ring while the submit and drain threads
    run; both endings of multifm/rtl_sdr_if.c - the read returns by itself, or SIGINT -> rtlsdr_cancel_async from the main
    thread while the callback is active.  ThreadSanitizer must have nothing to report."""
    import signal
    import time
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    exe = tmp_path / "multifm_tsan"
    srcs = [os.path.join(host, f) for f in ("mfm_tsl.c", "mfm_config.c", "mfm_receiver.c", "mfm_file_if.c", "mfm_rtl_sdr_if.c",
                                            "multifm_main.c")]
    srcs.append(os.path.join(ROOT, "tests", "hoststub", "stub_group.c"))
    r = subprocess.run(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=thread", "-D_GNU_SOURCE", "-I" + host,
                        "-I" + os.path.join(ROOT, "include"), "-o", str(exe)] + srcs + ["-lpthread", "-lm", "-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stderr
    fake_dir = tmp_path / "fakelib"
    _build_fake_rtlsdr(fake_dir)
    n = 16 * 32 * 512 // 2 * 6 + 1234          # six transfers and a bit
    cap = tmp_path / "rtl.u8"
    cap.write_bytes(np.random.RandomState(3).randint(0, 256, size=2 * n).astype(np.uint8).tobytes())
    outs = []
    cfg = {"device": {"type": "rtlsdr", "deviceIndex": 0, "dBGainLNA": 20.0}, "sampleRateHz": 1200000, "centerFreqHz": 152000000,
           "nrSampBufs": 8, "decimationFactor": 25, "lpfTaps": [0.01] * 50, "channels": []}
    for i, f in enumerate((-320000, 125000, 0)):
        o = tmp_path / f"ch{i}.pcm"
        o.write_bytes(b"")
        cfg["channels"].append({"outFifo": str(o), "chanCenterFreq": 152000000 + f})
        outs.append(o)
    if ending == "file_eof":
        # (the same driver on the file front end, 3 000 buffers of 4 096 samples: the reader outruns the device double's consumer)
        big = tmp_path / "cap.cs8"
        big.write_bytes(np.random.RandomState(4).randint(0, 256, size=2 * (4096 * 3000 + 77)).astype(np.uint8).tobytes())
        cfg["device"] = {"type": "file", "filename": str(big), "fileFormat": "cs8"}
    cj = tmp_path / "cfg.json"
    cj.write_text(json.dumps(cfg))
    log = tmp_path / "rtl.log"
    env = dict({k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}, LD_LIBRARY_PATH=str(fake_dir), FAKE_RTLSDR_FILE=str(cap),
               FAKE_RTLSDR_LOG=str(log), TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    if ending == "file_eof":
        r = subprocess.run([str(exe), str(cj)], capture_output=True, text=True, timeout=300, env=env)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
        m = re.search(r"INGEST-SUMMARY (\d+) sample buffers delivered, (\d+) submitted", r.stderr)
        assert r.returncode == 0 and m and int(m.group(1)) == int(m.group(2)) == 3001, (r.returncode, r.stderr[-2000:])
        return
    if ending == "reader_returns":
        r = subprocess.run([str(exe), str(cj)], capture_output=True, text=True, timeout=180, env=dict(env, FAKE_RTLSDR_EOF_RETURNS="1"))
        err, rc = r.stderr, r.returncode
    else:
        p = subprocess.Popen([str(exe), str(cj)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        deadline = time.time() + 120
        while time.time() < deadline and p.poll() is None and not (log.exists() and "read_async" in log.read_text()):
            time.sleep(0.05)
        time.sleep(0.3)                        # the callback is delivering (the capture is replayed again and again)
        assert p.poll() is None, p.stderr.read()[-3000:]
        p.send_signal(signal.SIGINT)
        _, err = p.communicate(timeout=120)
        rc = p.returncode
    assert "ThreadSanitizer" not in err, err[-4000:]
    assert rc == 0, (rc, err[-3000:])
    # (the device double takes the blocks and gives nothing back: no PCM is written; what counts is what the threads did)
    m = re.search(r"INGEST-SUMMARY (\d+) sample buffers delivered, (\d+) submitted", err)
    assert m and int(m.group(1)) == int(m.group(2)) >= 6, err[-2000:]
    names = [ln.split()[0] for ln in log.read_text().splitlines()]
    assert names[-1] == "close" and ("cancel_async" in names) == (ending == "sigint"), names


REF_FILE_IF = "/root/reference/multifm/file_if.c"


@pytest.mark.skipif(not os.path.exists(REF_FILE_IF), reason="reference tree not present (build container only)")
def test_reference_file_front_end_compiles_and_links_unchanged(tmp_path):
    """"file_if / rtl_sdr_if-style front ends compile unchanged against it".  The reference's
    own multifm/file_if.c (as it lies in /root/reference, nothing copied) is compiled against the compat include tree
    of the host library (tsl-sdr_amd/host/compat: <multifm/receiver.h>, <filter/sample_buf.h>, <config/engine.h>,
    <tsl/...> redirect to mfm_receiver.h / mfm_config.h / mfm_tsl.h) and linked with libmfm_host.so and this repo's
    multifm driver: its file_worker_thread_new then replaces the library's own."""
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    obj = tmp_path / "ref_file_if.o"
    cmd = ["gcc", "-std=gnu11", "-O2", "-D_GNU_SOURCE", "-Wall", "-Werror=implicit-function-declaration",
           "-I" + os.path.join(host, "compat"), "-I" + host, "-I" + os.path.join(ROOT, "include"), "-I/root/reference",
           "-c", "-o", str(obj), REF_FILE_IF]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", str(obj)], capture_output=True, text=True).stdout
    assert " T file_worker_thread_new" in syms
    undefined = {ln.split()[-1] for ln in syms.splitlines() if " U " in ln}
    # everything the front end needs from the receiver side is exported by the host library
    exported = subprocess.run(["nm", "-D", "--defined-only", HOST_SO], capture_output=True, text=True).stdout
    for sym in ("receiver_init", "receiver_sample_buf_alloc", "receiver_sample_buf_deliver", "receiver_thread_running",
                "config_get", "config_get_string", "tsl_get_clock_monotonic"):
        assert sym in undefined and f" T {sym}" in exported, sym
    exe = tmp_path / "multifm_ref_fileif"
    r = subprocess.run(["gcc", "-o", str(exe), os.path.join(host, "build", "multifm_main.o"), str(obj),
                        "-L" + host, "-lmfm_host", "-L" + os.path.join(ROOT, "tsl-sdr_amd"), "-lmultifm_hip",
                        "-Wl,-rpath," + host, "-Wl,-rpath," + os.path.join(ROOT, "tsl-sdr_amd"), "-lpthread", "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert " T file_worker_thread_new" in subprocess.run(["nm", str(exe)], capture_output=True, text=True).stdout


def _compat_cc(out, src, *extra):
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    return ["gcc", "-std=gnu11", "-O2", "-D_GNU_SOURCE", "-Wall", "-Werror=implicit-function-declaration",
            "-I" + os.path.join(host, "compat"), "-I" + host, "-I" + os.path.join(ROOT, "include"),
            "-I" + os.path.join(ROOT, "tests", "hoststub", "rtlsdr"), *extra, "-c", "-o", str(out), str(src)]


def _link_with_host(exe, objs, *extra):
    host = os.path.join(ROOT, "tsl-sdr_amd", "host")
    return ["gcc", "-o", str(exe), *[str(o) for o in objs], "-L" + host, "-lmfm_host",
            "-L" + os.path.join(ROOT, "tsl-sdr_amd"), "-lmultifm_hip", "-Wl,-rpath," + host,
            "-Wl,-rpath," + os.path.join(ROOT, "tsl-sdr_amd"), *extra, "-lpthread", "-lm"]


def _build_fake_rtlsdr(dirpath):
    """tests/hoststub/rtlsdr/fake_rtlsdr.c as librtlsdr.so.0 in a directory of its own (found via LD_LIBRARY_PATH)"""
    os.makedirs(dirpath, exist_ok=True)
    so = os.path.join(str(dirpath), "librtlsdr.so.0")
    stub = os.path.join(ROOT, "tests", "hoststub", "rtlsdr")
    r = subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I" + stub, "-o", so,
                        os.path.join(stub, "fake_rtlsdr.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return so


def test_tsl_names_of_the_compat_tree(tmp_path):
    """list_*, work_queue_*, TCALLOC, CAL_CLEANUP / free_memory, TSL_ASSERT_PTR_BY_REF, app_init /
    app_sigint_catch / app_running - used the way the reference uses them, through the compat include names."""
    obj, exe = tmp_path / "t.o", tmp_path / "tsl_compat_test"
    r = subprocess.run(_compat_cc(obj, os.path.join(ROOT, "tests", "hoststub", "tsl_compat_test.c"), "-Werror"),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(_link_with_host(exe, [obj]), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ)  # (tools/sanitize_cpu.sh: the host library is then the ASan build and wants its runtime preloaded)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1]) == {"bad": 0}, r.stdout + r.stderr


REF_MULTIFM_DIR = "/root/reference/multifm"


@pytest.mark.skipif(not os.path.exists(REF_MULTIFM_DIR), reason="reference tree not present (build container only)")
def test_reference_driver_and_rtl_sdr_front_end_compile_and_link_unchanged(tmp_path):
    """north_star: "drops in behind the existing rtl_sdr_if/file_if front ends".  The reference's multifm/multifm.c (its
    main), multifm/rtl_sdr_if.c and multifm/file_if.c, as they lie in /root/reference and built with -DHAVE_RTLSDR,
    compile against tsl-sdr_amd/host/compat (+ a test-side declaration file of librtlsdr's API) without a warning about
    an undeclared name, link against libmfm_host.so (+ the test double of librtlsdr), and the resulting program runs the
    reference main's own error paths: usage with the device list, unknown device type, missing stanzas."""
    objs = []
    for name in ("multifm", "rtl_sdr_if", "file_if"):
        obj = tmp_path / f"ref_{name}.o"
        r = subprocess.run(_compat_cc(obj, os.path.join(REF_MULTIFM_DIR, name + ".c"), "-DHAVE_RTLSDR", "-I/root/reference"),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        objs.append(obj)
    syms = {n: subprocess.run(["nm", str(o)], capture_output=True, text=True).stdout for n, o in zip(("main", "rtl", "file"), objs)}
    assert " T main" in syms["main"] and " T rtl_sdr_worker_thread_new" in syms["rtl"] and " T file_worker_thread_new" in syms["file"]
    exported = subprocess.run(["nm", "-D", "--defined-only", HOST_SO], capture_output=True, text=True).stdout
    needed = set()
    for txt in syms.values():
        needed |= {ln.split()[-1] for ln in txt.splitlines() if " U " in ln}
    # what the three files need from TSL / the receiver, all of it exported by the host library
    for sym in ("app_init", "app_sigint_catch", "app_running", "config_new", "config_add", "config_delete", "config_get",
                "config_get_string", "config_get_integer", "config_get_float", "config_get_boolean", "receiver_init",
                "receiver_start", "receiver_cleanup", "receiver_set_mute", "receiver_sample_buf_alloc",
                "receiver_sample_buf_deliver", "receiver_thread_running", "tsl_get_clock_monotonic"):
        assert sym in needed and f" T {sym}" in exported, sym
    ours = {s for s in needed if f" T {s}" in exported}
    theirs = {s for s in needed if s.startswith("rtlsdr_")}
    libc = needed - ours - theirs - {"rtl_sdr_worker_thread_new", "file_worker_thread_new", "_GLOBAL_OFFSET_TABLE_"}
    assert not {s for s in libc if s.startswith(("tsl_", "config_", "receiver_", "app_", "work_queue_", "frame_", "list_"))}, libc
    fake_dir = tmp_path / "fakelib"
    _build_fake_rtlsdr(fake_dir)
    exe = tmp_path / "multifm_ref"
    r = subprocess.run(_link_with_host(exe, objs, "-L" + str(fake_dir), "-l:librtlsdr.so.0", "-Wl,-rpath," + str(fake_dir)),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("FAKE_RTLSDR_FILE",)}
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 1 and "usage:" in r.stderr and "NO-DEVS-FOUND" in r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, env=dict(env, FAKE_RTLSDR_FILE="/dev/null"))
    assert r.returncode == 1 and "DEVS-FOUND" in r.stderr and "Fake RTL2838" in r.stderr
    for cfg, ident in (({"device": {"type": "bogus"}}, "UNKNOWN-DEV-TYPE"), ({"sampleRateHz": 1}, "MALFORMED-CONFIG"),
                       ({"device": {"nope": 1}}, "MALFORMED-CONFIG")):
        cj = tmp_path / "c.json"
        cj.write_text(json.dumps(cfg))
        r = subprocess.run([str(exe), str(cj)], capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 1 and ident in r.stderr, (cfg, r.stderr)
    (tmp_path / "broken.json").write_text("{ not json")
    r = subprocess.run([str(exe), str(tmp_path / "broken.json")], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 1 and "MALFORMED-CONFIG" in r.stderr


def test_rtl_sdr_front_end_without_the_library_and_usage_listing(tmp_path):
    """multifm_amd: without librtlsdr a "rtlsdr" device is refused with the reference's message for a build without it
    (multifm/multifm.c:132-135); with the library (the test double) the usage text lists the devices (:57-77)."""
    cj = tmp_path / "c.json"
    cj.write_text(json.dumps({"device": {"type": "rtlsdr", "deviceIndex": 0}, "sampleRateHz": 1200000, "centerFreqHz": 152000000}))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "FAKE_RTLSDR_FILE")}
    r = subprocess.run([MULTIFM, str(cj)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "RTLSDR-NOT-SUPPORTED" in r.stderr, r.stderr
    fake_dir = tmp_path / "fakelib"
    _build_fake_rtlsdr(fake_dir)
    r = subprocess.run([MULTIFM], capture_output=True, text=True, timeout=60,
                       env=dict(env, LD_LIBRARY_PATH=str(fake_dir), FAKE_RTLSDR_FILE="/dev/null"))
    assert r.returncode != 0 and "usage:" in r.stderr and "Fake RTL2838" in r.stderr, r.stderr


def test_frame_pool_and_refcount_contract(host):
    """nrSampBufs frames; exhaustion fails instead of blocking (receiver.c:57-63 then drops and counts);
    a buffer returns to the pool when the last holder decrefs it (sample_buf.c:31-43)."""
    fa = C.c_void_p()
    host.frame_alloc_new.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]
    host.frame_alloc.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    host.frame_free.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    host.frame_alloc_nr_free.argtypes = [C.c_void_p]
    host.frame_alloc_nr_free.restype = C.c_size_t
    host.frame_alloc_delete.argtypes = [C.POINTER(C.c_void_p)]
    assert host.frame_alloc_new(C.byref(fa), 48 + 4096 * 4, 4) == 0
    frames = []
    for _ in range(4):
        f = C.c_void_p()
        assert host.frame_alloc(fa, C.byref(f)) == 0 and f.value % 64 == 0
        frames.append(f)
    extra = C.c_void_p()
    assert host.frame_alloc(fa, C.byref(extra)) < 0 and not extra.value
    assert host.frame_alloc_nr_free(fa) == 0
    for f in frames:
        assert host.frame_free(fa, C.byref(f)) == 0
    assert host.frame_alloc_nr_free(fa) == 4
    assert host.frame_alloc_delete(C.byref(fa)) == 0


def _run_multifm(tmp_path, pkg, fmt, iq16, raw_bytes, fs, decim, center, chans, taps_file, gains_db=None, gpu_unpack=None):
    cap = tmp_path / f"cap_{fmt}.bin"
    cap.write_bytes(raw_bytes)
    dev = {"type": "file", "filename": str(cap), "fileFormat": fmt}
    if gpu_unpack is not None:
        dev["gpuUnpack"] = gpu_unpack
    cfg = {"device": dev, "sampleRateHz": fs,
           "centerFreqHz": center, "nrSampBufs": 32, "decimationFactor": decim, "channels": []}
    outs = []
    for i, f in enumerate(chans):
        o = tmp_path / f"ch{i}_{fmt}.pcm"
        o.write_bytes(b"")  # the driver opens sinks O_WRONLY like the reference opens its FIFOs
        ch = {"outFifo": str(o), "chanCenterFreq": int(center + f)}
        if gains_db and gains_db[i] is not None:
            ch["dBGain"] = gains_db[i]
        if i == 0:
            q = tmp_path / f"ch{i}_{fmt}.iq"
            q.write_bytes(b"")
            ch["signalDebugFile"] = str(q)
        cfg["channels"].append(ch)
        outs.append(o)
    cj = tmp_path / f"cfg_{fmt}.json"
    cj.write_text(json.dumps(cfg))
    r = subprocess.run([MULTIFM, str(cj), taps_file], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    _run_multifm.last_stderr = r.stderr
    return [np.frombuffer(o.read_bytes(), dtype=np.int16) for o in outs], \
        np.frombuffer((tmp_path / f"ch0_{fmt}.iq").read_bytes(), dtype=np.int16).reshape(-1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,gpu_unpack", [("cs16", None), ("cs8", None), ("cu8", None), ("cs8", False), ("cu8", False)])
def test_multifm_driver_on_file_input(tmp_path, pkg, ora, fmt, gpu_unpack):
    """BASELINE configs[0]: etc/multifm_1ch.json values (fs 1.0 MS/s, D 40, channel at +112.5 kHz) plus a second
    channel with dBGain, fed from a file through file_if's three sample formats; the FIFO byte streams must be
    the oracle's PCM.  8-bit formats are widened on the GPU by default and on the host with
    "gpuUnpack": false; both must reproduce file_if.c, including the last sample of the odd-sized final read."""
    fs, decim, center = 1000000, 40, 929500000
    offs, gains_db = [112500, -200000], [None, 4.0]
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_1000k_128.json")
    taps = np.array(json.load(open(taps_file))["lpfTaps"])
    n = 4096 * 37 + 1233  # last buffer is a partial one with an odd number of samples
    if fmt == "cs16":
        iq = pkg.synth.synth_iq(n, fs, offs, seed=91)
        raw = iq.tobytes()
    else:
        rng = np.random.RandomState(92)
        b = rng.randint(0, 256, size=(n, 2)).astype(np.uint8)
        raw = b.tobytes()
        # file_if.c:66-157, one read of 4096 samples at a time
        code = 1 if fmt == "cs8" else 2
        iq = np.concatenate([ora.unpack_bytes(b[i:i + 4096], code) for i in range(0, n, 4096)]).reshape(-1, 2)
        assert fmt == "cs8" or iq[-1, 0] == np.int8(b[-1, 0])  # the quirk: no -127 on the very last sample
    pcm, q = _run_multifm(tmp_path, pkg, fmt, iq, raw, fs, decim, center, offs, taps_file, gains_db, gpu_unpack)
    gains = [1.0, 10.0 ** (4.0 / 10.0)]
    cre = np.stack([ora.make_taps(taps, o, fs, g)[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, o, fs, g)[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(o, fs, decim) for o in offs])
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, want_iq=True)
    for c in range(2):
        assert pcm[c].shape == ref[c].shape and np.array_equal(pcm[c], ref[c]), f"channel {c} PCM differs"
    assert np.array_equal(q, refq[0])


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_unpack,ending", [(None, "reader_returns"), (False, "reader_returns"), (None, "sigint")])
def test_multifm_driver_on_rtl_sdr_input(tmp_path, pkg, ora, gpu_unpack, ending):
    """The RTL-SDR front end (tsl-sdr_amd/host/mfm_rtl_sdr_if.c; multifm/rtl_sdr_if.c:87-157,308-479) end to end at the
    geometry of the reference's etc/pocsag_rtlsdr.json (1.2 MS/s, decimation 25, channels at -320 kHz with dBGain 4 and at
    -492 kHz): a test double of librtlsdr (tests/hoststub/rtlsdr) hands out a capture in default-sized transfers, the last
    one short.  PCM and filtered IQ must be the oracle's on ((u8 - 127) << 7) samples, whether the bytes are widened on
    the GPU (default) or on the host; the dongle must have been programmed in the reference's order; and with a reader
    that blocks until cancelled (a real dongle) SIGINT must still deliver every output before the process ends."""
    import signal
    import time
    fs, decim, center = 1200000, 25, 929500000
    offs, gains_db = [-320000, -492000], [4.0, None]
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_1200k_128.json")
    taps = np.array(json.load(open(taps_file))["lpfTaps"])
    n = 131072 * 5 + 4099
    rng = np.random.RandomState(17)
    b = rng.randint(0, 256, size=(n, 2)).astype(np.uint8)
    cap = tmp_path / "rtl.u8"
    cap.write_bytes(b.tobytes())
    iq = ora.unpack_bytes(b, 3).reshape(-1, 2)
    assert iq[0, 0] == (int(b[0, 0]) - 127) * 128
    fake_dir = tmp_path / "fakelib"
    _build_fake_rtlsdr(fake_dir)
    dev = {"type": "rtlsdr", "deviceIndex": 0, "dBGainLNA": 30.0, "ppmCorrection": 37, "iqDumpFile": str(tmp_path / "dump.u8")}
    if gpu_unpack is not None:
        dev["gpuUnpack"] = gpu_unpack
    cfg = {"device": dev, "sampleRateHz": fs, "centerFreqHz": center, "nrSampBufs": 16, "decimationFactor": decim, "channels": []}
    outs = []
    for i, f in enumerate(offs):
        o = tmp_path / f"ch{i}.pcm"
        o.write_bytes(b"")
        ch = {"outFifo": str(o), "chanCenterFreq": int(center + f)}
        if gains_db[i] is not None:
            ch["dBGain"] = gains_db[i]
        if i == 0:
            (tmp_path / "ch0.iq").write_bytes(b"")
            ch["signalDebugFile"] = str(tmp_path / "ch0.iq")
        cfg["channels"].append(ch)
        outs.append(o)
    cj = tmp_path / "cfg.json"
    cj.write_text(json.dumps(cfg))
    log = tmp_path / "rtl.log"
    env = dict({k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}, LD_LIBRARY_PATH=str(fake_dir),
               FAKE_RTLSDR_FILE=str(cap), FAKE_RTLSDR_LOG=str(log))
    n_out = (n - len(taps)) // decim + 1
    if ending == "reader_returns":
        r = subprocess.run([MULTIFM, str(cj), taps_file], capture_output=True, text=True, timeout=180,
                           env=dict(env, FAKE_RTLSDR_EOF_RETURNS="1"))
        assert r.returncode == 0, r.stderr[-3000:]
    else:
        p = subprocess.Popen([MULTIFM, str(cj), taps_file], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        deadline = time.time() + 150
        while time.time() < deadline and p.poll() is None and min(o.stat().st_size for o in outs) < 2 * n_out:
            time.sleep(0.05)
        assert p.poll() is None, p.stderr.read()[-3000:]
        p.send_signal(signal.SIGINT)
        _, err = p.communicate(timeout=60)
        assert p.returncode == 0, err[-3000:]
    calls = [ln.split() for ln in log.read_text().splitlines()]
    names = [c[0] for c in calls]
    # the reference's programming order (multifm/rtl_sdr_if.c:366-452)
    order = ["open", "set_sample_rate", "set_center_freq", "set_agc_mode", "set_tuner_gain_mode", "set_tuner_gain",
             "set_freq_correction", "reset_buffer", "read_async"]
    assert [x for x in names if x in order] == order, names
    assert ["set_sample_rate", str(fs)] in calls and ["set_center_freq", str(center)] in calls
    assert ["set_tuner_gain_mode", "1"] in calls and ["set_tuner_gain", "328"] in calls  # first supported gain >= 30.0 dB
    assert ["set_freq_correction", "37"] in calls and ["read_async", str(16 * 32 * 512)] in calls
    assert names[-1] == "close" and names.index("read_async_returned") < names.index("close")
    assert ("cancel_async" in names) == (ending == "sigint")
    assert (tmp_path / "dump.u8").read_bytes() == b.tobytes()  # iqDumpFile: the raw bytes, every transfer
    pcm = [np.frombuffer(o.read_bytes(), dtype=np.int16) for o in outs]
    q = np.frombuffer((tmp_path / "ch0.iq").read_bytes(), dtype=np.int16).reshape(-1, 2)
    gains = [10.0 ** (4.0 / 10.0), 1.0]
    cre = np.stack([ora.make_taps(taps, o, fs, g)[0] for o, g in zip(offs, gains)])
    cim = np.stack([ora.make_taps(taps, o, fs, g)[1] for o, g in zip(offs, gains)])
    incr = np.stack([ora.rot_incr(o, fs, decim) for o in offs])
    ref, refq = ora.run_channels(iq, cre, cim, incr, decim, want_iq=True)
    assert ref.shape[1] == n_out
    for c in range(2):
        assert pcm[c].shape == ref[c].shape and np.array_equal(pcm[c], ref[c]), f"channel {c} PCM differs"
    assert np.array_equal(q, refq[0])


def test_decoder_driver_refuses_loudly_without_a_gpu_and_unknown_protocols(tmp_path):
    """decoder_amd: -m AIS is not built (exit 1, UNKNOWN-PROTOCOL-TYPE); without a device the FLEX (default) and POCSAG
    paths stop at the resampler with a message instead of falling back to anything on the CPU"""
    import json
    import subprocess
    tool = os.path.join(os.path.dirname(HOST_SO), "decoder_amd")
    if not os.path.exists(tool):
        pytest.fail(f"{tool} missing: run make -C tsl-sdr_amd")
    (tmp_path / "f.json").write_text(json.dumps({"lpfCoeffs": [0.25, 0.5, 0.25]}))
    (tmp_path / "in.pcm").write_bytes(np.zeros(4096, np.int16).tobytes())
    base = [tool, "-I", "1", "-D", "1", "-S", "16000", "-F", str(tmp_path / "f.json"), "-f", "929612500"]
    r = subprocess.run(base + ["-m", "AIS", str(tmp_path / "in.pcm")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "UNKNOWN-PROTOCOL-TYPE" in r.stderr
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    for proto in ([], ["-m", "FLEX"], ["-m", "POCSAG"]):
        r = subprocess.run(base + proto + [str(tmp_path / "in.pcm")], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "NO-RESAMPLER" in r.stderr, r.stderr[-500:]


def test_group_push_is_all_or_nothing_and_fetch_never_sees_half_a_push(tmp_path):
    """The ORDER of a device group's push and fetch (tsl-sdr_amd/csrc/mfm_group_seq.h, the template mfm_group.hip
    instantiates over its engines) on fake shards, under ThreadSanitizer: a shard that has no room, refuses its input
    buffer, a root whose copy fails, a collective that fails, a submit that fails after two shards took the block -
    nothing is staged or submitted where the reference's all-or-nothing delivery (multifm/receiver.c:78-98) forbids it, a
    half-submitted block makes every later call fail instead of letting the shards drift apart, and a consumer thread
    polling fetch while the producer sits between two submits only ever sees "nothing yet" or a whole block (round 2's
    "shards out of step" failure of the receiver)."""
    exe = tmp_path / "group_seq_test"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread", "-o", str(exe),
                        os.path.join(ROOT, "tests", "hoststub", "group_seq_test.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}  # tools/sanitize_cpu.sh preloads ASan: not into a TSan binary
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "all checks held" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
def test_multifm_driver_soak_ten_thousand_buffers(tmp_path, pkg, ora):
    """The C host end to end at length: 10 000 sample_bufs of a cs8 capture (41 M samples) through file_if -> receiver ->
    submit / drain threads -> 64 demod_thread sinks at the 2.4 MS/s / D = 96 plan.  Every sink must hold exactly the
    stream's outputs; six channels are compared with the oracle over three windows of 2 048 outputs each (first, middle,
    last), the first one also against the oracle's plain stream run to pin the numbering."""
    fs, decim, center = 2400000, 96, 152000000
    taps_file = os.path.join(ROOT, "etc", "lpf_25khz_2400k_128.json")
    if not os.path.exists(taps_file):
        pytest.skip("no 2.4 MS/s tap file in etc/")
    taps = np.array(json.load(open(taps_file))["lpfTaps"])
    offs = [int(o) for o in pkg.synth.channel_offsets(64, fs)]
    n = 4096 * 10000 + 777
    rng = np.random.RandomState(5)
    b = rng.randint(0, 256, size=(n, 2)).astype(np.uint8)
    iq = ora.unpack_bytes(b, 1).reshape(-1, 2)
    pcm, _ = _run_multifm(tmp_path, pkg, "cs8", iq, b.tobytes(), fs, decim, center, offs, taps_file)
    n_out = (n - len(taps)) // decim + 1
    # the file reader outruns the device: the pool's frames come in address order and runs of neighbours went to the device as
    # one strided copy command each (round 5) - far fewer commands than buffers
    import re
    m = re.search(r"INGEST-SUMMARY (\d+) sample buffers delivered, (\d+) submitted in (\d+) copy commands, (\d+) requests found", _run_multifm.last_stderr)
    assert m, _run_multifm.last_stderr[-1500:]
    assert int(m.group(1)) == int(m.group(2)) == 10001 and int(m.group(3)) <= 10001, m.groups()
    if int(m.group(4)) == 0:
        # (with the pool never empty; a file reader that finds it empty waits and then gets single frames as they come back -
        # e.g. under `pytest -n 4` on a shared box - and nothing neighbours anything)
        assert int(m.group(3)) < 5000, m.groups()
    for c in range(64):
        assert pcm[c].size == n_out, (c, pcm[c].size, n_out)
    for c in (0, 7, 8, 31, 40, 63):
        cre, cim = ora.make_taps(taps, offs[c], fs, 1.0)
        incr = ora.rot_incr(offs[c], fs, decim)
        head, _ = ora.run_channels(iq[:decim * 2100 + len(taps)], cre[None], cim[None], incr[None], decim)
        assert np.array_equal(pcm[c][:2100], head[0][:2100]), f"channel {c}: head differs"
        for w0 in (1, n_out // 2, n_out - 2048):
            want = ora.window_pcm(iq, cre, cim, decim, incr, 0, w0, 2048)
            assert np.array_equal(pcm[c][w0:w0 + 2048], want), f"channel {c}: window at {w0} differs"
