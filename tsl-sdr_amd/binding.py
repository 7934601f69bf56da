"""ctypes binding of include/multifm_hip.h (libmultifm_hip.so).

Used by tests/, bench.py and __graft_entry__.py.  It is a thin mirror of the C ABI: one Python
method per entry point, errors raised as MfmError carrying the library's message.  There is no
fallback: if the shared library is missing, importing the engine fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFM_LIB") or os.path.join(_HERE, "libmultifm_hip.so")

MFM_OK, MFM_E_INVAL, MFM_E_NOMEM, MFM_E_BUSY, MFM_E_DEVICE, MFM_E_STATE, MFM_E_DONE = 0, -1, -2, -3, -4, -5, -6
MFM_ABI_VERSION = 4
MFM_F_DEVICE_ONLY = 0x1
MFM_F_TIMING = 0x2
MFM_F_FORCE_DOT2 = 0x4
MFM_F_FORCE_MFMA_V1 = 0x8
MFM_F_WIDEN_8BIT = 0x10
MFM_F_TIMING_SPARSE = 0x20
MFM_F_GROUP_SHARED_DEVICE = 0x40
MFM_F_STREAM_TAPS = 0x80
MFM_F_GATHER = 0x100
MFM_F_OVERLAP = 0x200
MFM_F_V3L_ONE_ROW_BLOCK = 0x400
MFM_F_SLICE_128 = 0x800
MFM_F_SLICE_64 = 0x1000
MFM_F_PCM_WRITE_BACK = 0x2000
MFM_RCP_TABLE_HASH_GFX950 = 0x706D94BC005BCC1A  # include/multifm_hip.h
MFM_IN_CS16, MFM_IN_CS8, MFM_IN_CU8, MFM_IN_RTLSDR_U8 = 0, 1, 2, 3

# every symbol include/multifm_hip.h declares (tests check the library exports each one)
ABI_SYMBOLS = [
    "mfm_engine_input_bytes", "mfm_engine_input_bytes_cfg", "mfm_engine_flush", "mfm_engine_replay", "mfm_group_flush",
    "mfm_engine_last_launch_input", "mfm_engine_seek", "mfm_devtest_discriminate", "mfm_devtest_rcp_table",
    "mfm_host_alloc", "mfm_host_free", "mfm_engine_push_pinned", "mfm_engine_copy_done", "mfm_engine_copy_wait",
    "mfm_group_push_pinned", "mfm_group_copy_done", "mfm_group_copy_wait", "mfm_group_replay_pinned",
    "mfm_engine_create", "mfm_engine_destroy", "mfm_engine_add_channel",
    "mfm_engine_add_channel_q14", "mfm_engine_get_channel", "mfm_engine_commit", "mfm_engine_acquire_input",
    "mfm_engine_acquire_input_bytes",
    "mfm_engine_submit", "mfm_engine_push", "mfm_engine_push_bytes", "mfm_engine_fetch", "mfm_engine_release",
    "mfm_engine_last_output_device", "mfm_engine_sync", "mfm_engine_reset", "mfm_engine_get_stats",
    "mfm_engine_stream", "mfm_engine_get_launch_ms", "mfm_engine_get_launch_cycles", "mfm_group_acquire_input", "mfm_group_submit", "mfm_group_shard_engine", "mfm_link_probe", "mfm_link_probe_runs", "mfm_engine_push_pinned_run", "mfm_group_push_pinned_run", "mfm_engine_input_room", "mfm_group_replay_arena", "mfm_strerror", "mfm_last_error", "mfm_hosttwin_discriminate", "mfm_hosttwin_discriminate_batch", "mfm_hosttwin_r14",
    "mfm_hosttwin_pcm_range", "mfm_hosttwin_atan_table", "mfm_hosttwin_atan_table_ok",
    "mfm_resampler_create", "mfm_resampler_destroy", "mfm_resampler_max_out", "mfm_resampler_process_device",
    "mfm_resampler_process_host", "mfm_resampler_process_host_to_device",
    "mfm_pocsag_create", "mfm_pocsag_destroy", "mfm_pocsag_process_device", "mfm_pocsag_process_host",
    "mfm_pocsag_fetch_events", "mfm_bch3121_decode_device", "mfm_bch3121_decode_host", "mfm_hosttwin_bch3121_decode",
    "mfm_f32_create", "mfm_f32_add_channel", "mfm_f32_commit", "mfm_f32_destroy", "mfm_f32_max_out",
    "mfm_f32_process_device", "mfm_f32_process_host",
    "mfm_shard_range", "mfm_group_create", "mfm_group_destroy", "mfm_group_add_channel", "mfm_group_commit",
    "mfm_group_nr_shards", "mfm_group_shard_info", "mfm_group_push", "mfm_group_fetch", "mfm_group_release",
    "mfm_group_sync", "mfm_group_get_stats", "mfm_group_exchange_info", "mfm_group_exchange_detail", "mfm_group_rccl_library",
    "mfm_flex_create", "mfm_flex_destroy", "mfm_flex_process_device", "mfm_flex_process_host", "mfm_flex_fetch_events",
    "mfm_mm_create", "mfm_mm_destroy", "mfm_mm_max_decisions", "mfm_mm_process_device", "mfm_mm_process_host",
]

class ExchangeDetail(C.Structure):
    """struct mfm_exchange_detail"""
    _fields_ = [("device", C.c_int32), ("rccl_ranks", C.c_int32), ("pci_bus_id", C.c_char * 32), ("timed_exchanges", C.c_uint64),
                ("exchange_ms", C.c_double), ("timed_launches", C.c_uint64), ("kernel_ms", C.c_double), ("bound", C.c_uint32),
                ("reserved0", C.c_uint32)]


def rccl_library():
    """the RCCL file a device group of more than one GPU would use (mfm_group_rccl_library); raises MfmError when none loads"""
    lib = load_library()
    buf = C.create_string_buffer(1024)
    rc = lib.mfm_group_rccl_library(buf, len(buf))
    if rc < 0:
        raise MfmError(rc, "mfm_group_rccl_library", lib.mfm_last_error().decode())
    return buf.value.decode()


MFM_POCSAG_EV_SYNC_FOUND, MFM_POCSAG_EV_BATCH, MFM_POCSAG_EV_SYNC_LOST, MFM_POCSAG_EV_SYNC_KEPT = 1, 2, 3, 4


class MfmError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__(f"{what}: {detail} (code {code})")
        self.code = code


class EngineConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("sample_rate_hz", C.c_uint32),
                ("decimation", C.c_uint32), ("max_block_samples", C.c_uint32), ("flags", C.c_uint32),
                ("ext_input", C.c_void_p * 3), ("coalesce_samples", C.c_uint32), ("reserved", C.c_uint32)]


class GroupConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("nr_devices", C.c_uint32), ("devices", C.c_int32 * 16),
                ("sample_rate_hz", C.c_uint32), ("decimation", C.c_uint32), ("max_block_samples", C.c_uint32),
                ("flags", C.c_uint32), ("exchange", C.c_uint32), ("coalesce_samples", C.c_uint32),
                ("reserved", C.c_uint32)]


MFM_X_AUTO, MFM_X_RCCL, MFM_X_RCCL_ALLGATHER = 0, 1, 2


class Block(C.Structure):
    _fields_ = [("first_output", C.c_uint64), ("nr_outputs", C.c_size_t), ("stride", C.c_size_t),
                ("pcm", C.POINTER(C.c_int16)), ("iq", C.POINTER(C.c_int16))]


class Stats(C.Structure):
    _fields_ = [("samples_in", C.c_uint64), ("outputs", C.c_uint64), ("launches", C.c_uint64),
                ("kernel_ms", C.c_double), ("nr_channels", C.c_uint32), ("nr_taps", C.c_uint32),
                ("outputs_per_tile", C.c_uint32), ("lds_bytes", C.c_uint32), ("grid_last", C.c_uint32),
                ("tail_samples", C.c_uint32), ("rot_table_entries", C.c_uint64),
                ("kernel_variant", C.c_uint32), ("pending_blocks", C.c_uint32),
                ("launches_8bit", C.c_uint64), ("timed_launches", C.c_uint64),
                ("rot_exact_channels", C.c_uint32), ("rot_fast_slices", C.c_uint32),
                ("k_steps", C.c_uint32), ("tap_hi_mask", C.c_uint32),
                ("taps_resident", C.c_uint32), ("slice_channels", C.c_uint32),
                ("submits", C.c_uint64), ("pending_samples", C.c_uint64)]


class PocsagConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("nr_channels", C.c_uint32),
                ("max_in_samples", C.c_uint32), ("max_events", C.c_uint32), ("flags", C.c_uint32)]


class PocsagEvent(C.Structure):
    _fields_ = [("type", C.c_uint32), ("baud", C.c_uint32), ("channel", C.c_uint32), ("aux", C.c_uint32),
                ("sample", C.c_uint64), ("nr_ok", C.c_uint32), ("fail_mask", C.c_uint32),
                ("raw", C.c_uint32 * 16), ("corrected", C.c_uint32 * 16)]


# numpy view of struct mfm_pocsag_event (160 bytes)
POCSAG_EVENT_DTYPE = np.dtype([("type", "<u4"), ("baud", "<u4"), ("channel", "<u4"), ("aux", "<u4"), ("sample", "<u8"),
                               ("nr_ok", "<u4"), ("fail_mask", "<u4"), ("raw", "<u4", (16,)), ("corrected", "<u4", (16,))])


class FlexConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("nr_channels", C.c_uint32),
                ("max_in_samples", C.c_uint32), ("max_events", C.c_uint32), ("flags", C.c_uint32)]


# numpy views of struct mfm_flex_event (88 bytes) and struct mfm_flex_frame_words
FLEX_EVENT_DTYPE = np.dtype([("type", "<u4"), ("channel", "<u4"), ("sample", "<u8"), ("sync_sample", "<u8"),
                             ("coding", "<u4"), ("baud", "<u4"), ("eye", "<u4"), ("a", "<u4"), ("b", "<u4"),
                             ("inv_a", "<u4"), ("fiw_raw", "<u4"), ("fiw", "<u4"), ("fiw_rc", "<u4"),
                             ("sample_range", "<i4"), ("sample_delta", "<i4"), ("cycle", "<u4"), ("frame", "<u4"),
                             ("frame_index", "<u4"), ("nr_phases", "<u4"), ("reserved", "<u4")])
FLEX_FRAME_DTYPE = np.dtype([("words", "<u4", (4, 88))])
MFM_FLEX_EV_FRAME, MFM_FLEX_EV_BAD_BAUD, MFM_FLEX_EV_BAD_FIW = 1, 2, 3


class ResamplerConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("nr_channels", C.c_uint32),
                ("interpolate", C.c_uint32), ("decimate", C.c_uint32), ("max_in_samples", C.c_uint32),
                ("invert", C.c_uint32), ("dc_block", C.c_uint32), ("dc_pole", C.c_double), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


MFM_RS_FORCE_DOT2 = 1


class F32Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("sample_rate_hz", C.c_uint32),
                ("decimation", C.c_uint32), ("max_block_samples", C.c_uint32), ("flags", C.c_uint32)]


class F32Block(C.Structure):
    _fields_ = [("d_pcm_f32", C.c_void_p), ("d_pcm_i16", C.c_void_p), ("d_iq_f32", C.c_void_p),
                ("stride", C.c_size_t), ("nr_out", C.c_size_t), ("nr_channels", C.c_uint32), ("reserved", C.c_uint32)]


MFM_F32_WANT_IQ = 1
MFM_F32_PACKED_FMA = 2
MFM_F32_TILE_KERNEL = 4


class MmConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("nr_channels", C.c_uint32),
                ("max_in_samples", C.c_uint32), ("kw", C.c_float), ("km", C.c_float), ("samples_per_bit", C.c_float),
                ("error_min", C.c_float), ("error_max", C.c_float)]

_lib = None


def load_library():
    """dlopen libmultifm_hip.so (raises if it was not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make -C tsl-sdr_amd` (hipcc, gfx950). "
                          "There is no CPU fallback for the multifm engine.")
    lib = C.CDLL(LIB_PATH)
    vp, i16p = C.c_void_p, C.POINTER(C.c_int16)
    lib.mfm_engine_input_bytes.restype = C.c_size_t
    lib.mfm_engine_input_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.mfm_engine_input_bytes_cfg.restype = C.c_size_t
    lib.mfm_engine_input_bytes_cfg.argtypes = [C.POINTER(EngineConfig), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.mfm_engine_flush.argtypes = [vp]
    lib.mfm_engine_replay.argtypes = [vp, C.c_size_t, C.c_size_t]
    lib.mfm_engine_seek.argtypes = [vp, C.c_uint64]
    lib.mfm_host_alloc.argtypes = [C.c_size_t]
    lib.mfm_host_alloc.restype = vp
    lib.mfm_host_free.argtypes = [vp]
    lib.mfm_host_free.restype = None
    lib.mfm_engine_push_pinned.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    lib.mfm_engine_copy_done.argtypes = [vp, C.c_uint64]
    lib.mfm_engine_copy_wait.argtypes = [vp, C.c_uint64]
    lib.mfm_group_push_pinned.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)]
    lib.mfm_group_copy_done.argtypes = [vp, C.c_uint64]
    lib.mfm_group_replay_pinned.argtypes = [vp, C.POINTER(vp), C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_uint64)]
    lib.mfm_group_copy_wait.argtypes = [vp, C.c_uint64]
    lib.mfm_engine_last_launch_input.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    lib.mfm_group_flush.argtypes = [vp]
    lib.mfm_engine_create.argtypes = [C.POINTER(vp), C.POINTER(EngineConfig)]
    lib.mfm_engine_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_engine_destroy.restype = None
    lib.mfm_engine_add_channel.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int]
    lib.mfm_engine_add_channel_q14.argtypes = [vp, i16p, i16p, C.c_size_t, C.c_int16, C.c_int16, C.c_int]
    lib.mfm_engine_get_channel.argtypes = [vp, C.c_uint32, i16p, i16p, i16p]
    lib.mfm_engine_commit.argtypes = [vp]
    lib.mfm_engine_acquire_input.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.mfm_engine_acquire_input_bytes.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.mfm_engine_submit.argtypes = [vp, C.c_size_t, vp, C.c_int]
    lib.mfm_engine_push.argtypes = [vp, i16p, C.c_size_t]
    lib.mfm_engine_push_bytes.argtypes = [vp, vp, C.c_size_t, C.c_int]
    lib.mfm_engine_fetch.argtypes = [vp, C.POINTER(Block)]
    lib.mfm_engine_release.argtypes = [vp]
    lib.mfm_engine_last_output_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t),
                                                  C.POINTER(C.c_size_t), C.POINTER(vp)]
    lib.mfm_engine_sync.argtypes = [vp]
    lib.mfm_engine_reset.argtypes = [vp]
    lib.mfm_engine_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.mfm_engine_stream.argtypes = [vp]
    lib.mfm_shard_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mfm_shard_range.restype = None
    lib.mfm_group_create.argtypes = [C.POINTER(vp), C.POINTER(GroupConfig)]
    lib.mfm_group_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_group_destroy.restype = None
    lib.mfm_group_add_channel.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_int]
    lib.mfm_group_commit.argtypes = [vp]
    lib.mfm_group_nr_shards.argtypes = [vp]
    lib.mfm_group_shard_info.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    lib.mfm_group_push.argtypes = [vp, vp, C.c_size_t, C.c_int]
    lib.mfm_group_fetch.argtypes = [vp, C.POINTER(Block)]
    lib.mfm_group_release.argtypes = [vp]
    lib.mfm_group_sync.argtypes = [vp]
    lib.mfm_group_get_stats.argtypes = [vp, C.c_uint32, C.POINTER(Stats)]
    lib.mfm_group_exchange_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.mfm_group_acquire_input.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.mfm_group_submit.argtypes = [vp, C.c_size_t]
    lib.mfm_group_shard_engine.argtypes = [vp, C.c_uint32]
    lib.mfm_group_shard_engine.restype = vp
    lib.mfm_link_probe.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mfm_engine_push_pinned_run.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    lib.mfm_group_push_pinned_run.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    lib.mfm_engine_input_room.argtypes = [vp]
    lib.mfm_engine_input_room.restype = C.c_size_t
    lib.mfm_group_replay_arena.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
    lib.mfm_link_probe_runs.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double)]
    lib.mfm_engine_get_launch_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t]
    lib.mfm_engine_get_launch_ms.restype = C.c_size_t
    lib.mfm_engine_get_launch_cycles.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t]
    lib.mfm_engine_get_launch_cycles.restype = C.c_size_t
    lib.mfm_engine_stream.restype = vp
    if hasattr(lib, "mfm_group_exchange_detail"):   # (an older library under MFM_LIB, tools/exp/ab.py: same-box A/B against it)
        lib.mfm_group_exchange_detail.argtypes = [vp, C.c_uint32, vp]
        lib.mfm_group_rccl_library.argtypes = [C.c_char_p, C.c_size_t]
    lib.mfm_strerror.argtypes = [C.c_int]
    lib.mfm_strerror.restype = C.c_char_p
    lib.mfm_last_error.restype = C.c_char_p
    lib.mfm_hosttwin_discriminate.argtypes = [C.c_int32, C.c_int32]
    lib.mfm_hosttwin_discriminate.restype = C.c_int32
    lib.mfm_hosttwin_discriminate_batch.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, i16p]
    lib.mfm_hosttwin_discriminate_batch.restype = None
    lib.mfm_devtest_discriminate.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, i16p, C.c_int]
    lib.mfm_devtest_rcp_table.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.mfm_hosttwin_r14.argtypes = [C.c_int32]
    lib.mfm_hosttwin_r14.restype = C.c_int16
    lib.mfm_hosttwin_pcm_range.argtypes = [C.c_uint32, C.c_uint32, i16p]
    lib.mfm_hosttwin_pcm_range.restype = None
    lib.mfm_hosttwin_atan_table.argtypes = [C.POINTER(C.c_float)]
    lib.mfm_hosttwin_atan_table.restype = None
    lib.mfm_hosttwin_atan_table_ok.restype = C.c_int
    u32p = C.POINTER(C.c_uint32)
    lib.mfm_pocsag_create.argtypes = [C.POINTER(vp), C.POINTER(PocsagConfig)]
    lib.mfm_pocsag_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_pocsag_destroy.restype = None
    lib.mfm_pocsag_process_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    lib.mfm_pocsag_process_host.argtypes = [vp, i16p, C.c_size_t, C.c_size_t]
    lib.mfm_pocsag_fetch_events.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.mfm_flex_create.argtypes = [C.POINTER(vp), C.POINTER(FlexConfig)]
    lib.mfm_flex_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_flex_destroy.restype = None
    lib.mfm_flex_process_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    lib.mfm_flex_process_host.argtypes = [vp, i16p, C.c_size_t, C.c_size_t]
    lib.mfm_flex_fetch_events.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.mfm_bch3121_decode_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    lib.mfm_bch3121_decode_host.argtypes = [u32p, C.POINTER(C.c_uint8), C.c_size_t, C.c_int]
    lib.mfm_hosttwin_bch3121_decode.argtypes = [u32p]
    lib.mfm_resampler_create.argtypes = [C.POINTER(vp), C.POINTER(ResamplerConfig), i16p, C.c_size_t]
    lib.mfm_resampler_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_resampler_destroy.restype = None
    lib.mfm_resampler_max_out.argtypes = [vp]
    lib.mfm_resampler_max_out.restype = C.c_size_t
    lib.mfm_resampler_process_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.POINTER(vp),
                                                 C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.mfm_resampler_process_host.argtypes = [vp, i16p, C.c_size_t, C.c_size_t, i16p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.mfm_resampler_process_host_to_device.argtypes = [vp, i16p, C.c_size_t, C.c_size_t, vp, C.POINTER(vp),
                                                         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    f32p = C.POINTER(C.c_float)
    lib.mfm_f32_create.argtypes = [C.POINTER(vp), C.POINTER(F32Config)]
    lib.mfm_f32_add_channel.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_size_t, C.c_double]
    lib.mfm_f32_commit.argtypes = [vp]
    lib.mfm_f32_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_f32_destroy.restype = None
    lib.mfm_f32_max_out.argtypes = [vp]
    lib.mfm_f32_max_out.restype = C.c_size_t
    lib.mfm_f32_process_device.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(F32Block)]
    lib.mfm_f32_process_host.argtypes = [vp, f32p, C.c_size_t, f32p, i16p, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.mfm_mm_create.argtypes = [C.POINTER(vp), C.POINTER(MmConfig)]
    lib.mfm_mm_destroy.argtypes = [C.POINTER(vp)]
    lib.mfm_mm_destroy.restype = None
    lib.mfm_mm_max_decisions.argtypes = [vp]
    lib.mfm_mm_max_decisions.restype = C.c_size_t
    lib.mfm_mm_process_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.POINTER(vp), C.POINTER(C.c_size_t),
                                          C.POINTER(vp)]
    lib.mfm_mm_process_host.argtypes = [vp, i16p, C.c_size_t, C.c_size_t, i16p, C.c_size_t, u32p]
    _lib = lib
    return lib


def _i16p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


class Engine:
    """One multifm channel engine (struct mfm_engine) on one GPU."""

    def __init__(self, sample_rate_hz, decimation, max_block_samples, device=0, flags=0, ext_input=None,
                 coalesce_samples=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = EngineConfig()
        cfg.abi_version = MFM_ABI_VERSION
        cfg.device = device
        cfg.sample_rate_hz = sample_rate_hz
        cfg.decimation = decimation
        cfg.max_block_samples = max_block_samples
        cfg.flags = flags
        cfg.coalesce_samples = coalesce_samples
        if ext_input is not None:
            for i, ptr in enumerate(ext_input):
                cfg.ext_input[i] = ptr
        self.sample_rate_hz, self.decimation, self.max_block_samples = sample_rate_hz, decimation, max_block_samples
        self.nr_taps = 0
        self.nr_channels = 0
        self._chk(self.lib.mfm_engine_create(C.byref(self.h), C.byref(cfg)), "mfm_engine_create")

    def _chk(self, rc, what):
        if rc < 0:
            raise MfmError(rc, what, self.lib.mfm_last_error().decode() or self.lib.mfm_strerror(rc).decode())
        return rc

    def close(self):
        if self.h and not getattr(self, "_borrowed", False):  # (a shard engine of a Group belongs to the group)
            self.lib.mfm_engine_destroy(C.byref(self.h))
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- channel set -------------------------------------------------------------------
    def add_channel(self, offset_hz, lpf_taps, gain=1.0, want_iq=False):
        taps = np.ascontiguousarray(lpf_taps, dtype=np.float64)
        idx = self._chk(self.lib.mfm_engine_add_channel(self.h, int(offset_hz),
                                                        taps.ctypes.data_as(C.POINTER(C.c_double)), taps.size,
                                                        float(gain), int(want_iq)), "mfm_engine_add_channel")
        self.nr_taps = taps.size
        self.nr_channels = idx + 1
        return idx

    def add_channel_q14(self, coeff_re, coeff_im, incr, want_iq=False):
        cre = np.ascontiguousarray(coeff_re, dtype=np.int16)
        cim = np.ascontiguousarray(coeff_im, dtype=np.int16)
        idx = self._chk(self.lib.mfm_engine_add_channel_q14(self.h, _i16p(cre), _i16p(cim), cre.size,
                                                            int(incr[0]), int(incr[1]), int(want_iq)),
                        "mfm_engine_add_channel_q14")
        self.nr_taps = cre.size
        self.nr_channels = idx + 1
        return idx

    def get_channel(self, chan):
        cre = np.zeros(self.nr_taps, dtype=np.int16)
        cim = np.zeros(self.nr_taps, dtype=np.int16)
        incr = np.zeros(2, dtype=np.int16)
        self._chk(self.lib.mfm_engine_get_channel(self.h, chan, _i16p(cre), _i16p(cim), _i16p(incr)),
                  "mfm_engine_get_channel")
        return cre, cim, incr

    def commit(self):
        self._chk(self.lib.mfm_engine_commit(self.h), "mfm_engine_commit")

    # -- data path ---------------------------------------------------------------------
    def acquire_input(self):
        ptr, cap = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.mfm_engine_acquire_input(self.h, C.byref(ptr), C.byref(cap)), "mfm_engine_acquire_input")
        return ptr.value, cap.value

    def acquire_input_bytes(self, fmt):
        """where the next block goes as 8-bit IQ bytes (two per sample); MfmError(MFM_E_STATE) when the engine cannot
        read that format as bytes now"""
        ptr, cap = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.mfm_engine_acquire_input_bytes(self.h, fmt, C.byref(ptr), C.byref(cap)),
                  "mfm_engine_acquire_input_bytes")
        return ptr.value, cap.value

    def submit(self, nr_samples, producer_stream=None, wait_producer=None):
        """producer_stream: raw hipStream_t (0 = legacy default stream).  By default the engine waits for
        it whenever one is given."""
        if wait_producer is None:
            wait_producer = producer_stream is not None
        self._chk(self.lib.mfm_engine_submit(self.h, nr_samples, C.c_void_p(producer_stream or 0),
                                             int(bool(wait_producer))), "mfm_engine_submit")

    def push(self, iq):
        """iq: int16 array of interleaved I,Q (2*n elements)."""
        a = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        return self.lib.mfm_engine_push(self.h, _i16p(a), a.size // 2)

    def push_bytes(self, raw, fmt):
        """raw: 8-bit IQ pairs as uint8/int8 (2*n elements) or int16 for MFM_IN_CS16; widened on the device."""
        a = np.ascontiguousarray(raw).reshape(-1)
        nr = a.size // 2
        return self.lib.mfm_engine_push_bytes(self.h, a.ctypes.data, nr, fmt)

    def fetch(self):
        """Oldest finished block as (first_output, pcm[C][n] copy, iq[C][n][2] copy or None); None when drained."""
        blk = Block()
        rc = self.lib.mfm_engine_fetch(self.h, C.byref(blk))
        if rc == MFM_E_DONE:
            return None
        self._chk(rc, "mfm_engine_fetch")
        n, stride, nch = blk.nr_outputs, blk.stride, self.nr_channels
        pcm = np.ctypeslib.as_array(blk.pcm, shape=(nch, stride))[:, :n].copy()
        iq = None
        if blk.iq:
            iq = np.ctypeslib.as_array(blk.iq, shape=(nch, stride, 2))[:, :n, :].copy()
        first = blk.first_output
        self._chk(self.lib.mfm_engine_release(self.h), "mfm_engine_release")
        return first, pcm, iq

    def run(self, iq, block_samples):
        """Push a whole stream in blocks, draining as needed.  Returns (pcm[C][N], iq[C][N][2] | None)."""
        a = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1, 2)
        pcm_parts, iq_parts = [], []

        def drain():
            while True:
                got = self.fetch()
                if got is None:
                    return
                pcm_parts.append(got[1])
                if got[2] is not None:
                    iq_parts.append(got[2])

        pos = 0
        while pos < a.shape[0]:
            n = min(block_samples, a.shape[0] - pos)
            rc = self.push(a[pos:pos + n])
            if rc == MFM_E_BUSY:
                drain()
                continue
            self._chk(rc, "mfm_engine_push")
            pos += n
        # a coalescing engine may hold accepted blocks it has not launched yet
        while self.flush() == MFM_E_BUSY:
            drain()
        drain()
        nch = self.nr_channels
        pcm = np.concatenate(pcm_parts, axis=1) if pcm_parts else np.zeros((nch, 0), np.int16)
        iqo = np.concatenate(iq_parts, axis=1) if iq_parts else None
        return pcm, iqo

    def last_output_device(self):
        p, st, n, q = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_void_p()
        self._chk(self.lib.mfm_engine_last_output_device(self.h, C.byref(p), C.byref(st), C.byref(n), C.byref(q)),
                  "mfm_engine_last_output_device")
        return p.value, st.value, n.value, q.value

    def flush(self):
        """launch what a coalescing engine has accepted and not launched; returns 0 or MFM_E_BUSY"""
        rc = self.lib.mfm_engine_flush(self.h)
        if rc == MFM_E_BUSY:
            return rc
        return self._chk(rc, "mfm_engine_flush")

    def replay(self, block_samples, nr_blocks):
        """nr_blocks x { acquire_input; submit(block_samples) } in C on whatever the input buffers hold"""
        self._chk(self.lib.mfm_engine_replay(self.h, block_samples, nr_blocks), "mfm_engine_replay")

    def last_launch_input(self):
        """(device address, samples, MFM_IN_* format) of what the most recent launch read"""
        p, n, f = C.c_void_p(), C.c_size_t(), C.c_int()
        self._chk(self.lib.mfm_engine_last_launch_input(self.h, C.byref(p), C.byref(n), C.byref(f)),
                  "mfm_engine_last_launch_input")
        return p.value, n.value, f.value

    def sync(self):
        self._chk(self.lib.mfm_engine_sync(self.h), "mfm_engine_sync")

    def reset(self):
        self._chk(self.lib.mfm_engine_reset(self.h), "mfm_engine_reset")

    def seek(self, outputs_before):
        """fresh history, rotators where outputs_before steps leave them, output numbering continues from there"""
        self._chk(self.lib.mfm_engine_seek(self.h, int(outputs_before)), "mfm_engine_seek")

    def stats(self):
        st = Stats()
        self._chk(self.lib.mfm_engine_get_stats(self.h, C.byref(st)), "mfm_engine_get_stats")
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def launch_ms(self, last=4096):
        """MFM_F_TIMING: durations (ms) of the most recent `last` launches, oldest first."""
        buf = np.zeros(last, np.float32)
        n = self.lib.mfm_engine_get_launch_ms(self.h, buf.ctypes.data_as(C.POINTER(C.c_float)), last)
        return buf[:n].copy()

    def launch_cycles(self, last=1024):
        """MFM_F_TIMING, second-generation kernels: (shader-clock ticks, 100 MHz reference ticks) of the most recent `last`
        launches, oldest first, as the kernel stamped them (0 where a launch left no stamp)."""
        a, b = np.zeros(last, np.uint64), np.zeros(last, np.uint64)
        n = self.lib.mfm_engine_get_launch_cycles(self.h, a.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   b.ctypes.data_as(C.POINTER(C.c_uint64)), last)
        return a[:n].copy(), b[:n].copy()

    @property
    def stream(self):
        return self.lib.mfm_engine_stream(self.h)


def shard_range(nr_channels, nr_shards, shard):
    """mfm_shard_range: (first, count) of a shard's contiguous channel range."""
    lib = load_library()
    lo, n = C.c_uint32(), C.c_uint32()
    lib.mfm_shard_range(nr_channels, nr_shards, shard, C.byref(lo), C.byref(n))
    return lo.value, n.value


class Group:
    """mfm_group_*: one channel set on several devices of a node (RCCL broadcast of every block)."""

    def __init__(self, sample_rate_hz, decimation, max_block_samples, devices=(0,), flags=0, exchange=MFM_X_AUTO,
                 coalesce_samples=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = GroupConfig()
        cfg.abi_version = MFM_ABI_VERSION
        cfg.nr_devices = len(devices)
        for i, d in enumerate(devices):
            cfg.devices[i] = d
        cfg.sample_rate_hz, cfg.decimation, cfg.max_block_samples = sample_rate_hz, decimation, max_block_samples
        cfg.flags, cfg.exchange = flags, exchange
        cfg.coalesce_samples = coalesce_samples
        rc = self.lib.mfm_group_create(C.byref(self.h), C.byref(cfg))
        if rc < 0:
            raise MfmError(rc, "mfm_group_create", self.lib.mfm_last_error().decode())
        self.nr_channels = 0
        self._decimation, self._sample_rate_hz, self._nr_taps = decimation, sample_rate_hz, 0

    def _chk(self, rc, what):
        if rc < 0:
            raise MfmError(rc, what, self.lib.mfm_last_error().decode() or self.lib.mfm_strerror(rc).decode())
        return rc

    def close(self):
        if self.h:
            self.lib.mfm_group_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_channel(self, offset_hz, lpf_taps, gain=1.0, want_iq=False):
        t = np.ascontiguousarray(lpf_taps, dtype=np.float64)
        self.nr_channels += 1
        self._nr_taps = t.size
        return self._chk(self.lib.mfm_group_add_channel(self.h, int(offset_hz), t.ctypes.data_as(C.POINTER(C.c_double)),
                                                        t.size, float(gain), int(want_iq)), "mfm_group_add_channel")

    def commit(self):
        self._chk(self.lib.mfm_group_commit(self.h), "mfm_group_commit")
        self.nr_shards = self._chk(self.lib.mfm_group_nr_shards(self.h), "mfm_group_nr_shards")

    def shard_info(self, shard):
        lo, n, dev = C.c_uint32(), C.c_uint32(), C.c_int32()
        self._chk(self.lib.mfm_group_shard_info(self.h, shard, C.byref(lo), C.byref(n), C.byref(dev)), "mfm_group_shard_info")
        return lo.value, n.value, dev.value

    def push(self, data, fmt=MFM_IN_CS16):
        """returns 0 or MFM_E_BUSY; data: int16 [n, 2] (cs16) or uint8 [n, 2]"""
        a = np.ascontiguousarray(data)
        rc = self.lib.mfm_group_push(self.h, a.ctypes.data, a.shape[0], fmt)
        if rc == MFM_E_BUSY:
            return rc
        return self._chk(rc, "mfm_group_push")

    def fetch(self):
        """oldest finished block of all shards as one [nr_channels, n] array, or None"""
        blks = (Block * self.nr_shards)()
        rc = self.lib.mfm_group_fetch(self.h, blks)
        if rc == MFM_E_DONE:
            return None
        self._chk(rc, "mfm_group_fetch")
        parts = []
        for s in range(self.nr_shards):
            _, n, _ = self.shard_info(s)
            b = blks[s]
            arr = np.ctypeslib.as_array(b.pcm, shape=(n, b.stride))[:, :b.nr_outputs].copy()
            parts.append(arr)
        first = blks[0].first_output
        self._chk(self.lib.mfm_group_release(self.h), "mfm_group_release")
        return first, np.concatenate(parts, axis=0)

    def flush(self):
        """launch, on every shard, what has been pushed and not launched; returns 0 or MFM_E_BUSY"""
        rc = self.lib.mfm_group_flush(self.h)
        if rc == MFM_E_BUSY:
            return rc
        return self._chk(rc, "mfm_group_flush")

    def sync(self):
        self._chk(self.lib.mfm_group_sync(self.h), "mfm_group_sync")

    def stats(self, shard):
        st = Stats()
        self._chk(self.lib.mfm_group_get_stats(self.h, shard, C.byref(st)), "mfm_group_get_stats")
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def acquire_input(self):
        """(device address, capacity in samples) of where the next device-resident block goes: the root's input buffer"""
        p, cap = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.mfm_group_acquire_input(self.h, C.byref(p), C.byref(cap)), "mfm_group_acquire_input")
        return p.value, cap.value

    def submit(self, nr_samples):
        """the block at acquire_input()'s address: exchanged and submitted on every shard; returns 0 or MFM_E_BUSY"""
        rc = self.lib.mfm_group_submit(self.h, nr_samples)
        if rc == MFM_E_BUSY:
            return rc
        return self._chk(rc, "mfm_group_submit")

    def shard_engine(self, shard):
        """shard `shard`'s engine as an Engine object for the read-only calls (stats, launch_ms, launch_cycles,
        last_output_device, last_launch_input, get_channel); it does not own the handle"""
        h = self.lib.mfm_group_shard_engine(self.h, shard)
        if not h:
            raise MfmError(-1, "mfm_group_shard_engine", "no such shard")
        e = Engine.__new__(Engine)
        e.lib, e.h, e._borrowed = self.lib, C.c_void_p(h), True
        e.decimation, e.sample_rate_hz = self._decimation, self._sample_rate_hz
        e.nr_channels = self.shard_info(shard)[1]
        e.nr_taps = self._nr_taps
        return e

    def exchange_info(self):
        u, b, x = C.c_int(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.mfm_group_exchange_info(self.h, C.byref(u), C.byref(b), C.byref(x)), "mfm_group_exchange_info")
        return bool(u.value), b.value, x.value

    def exchange_detail(self, shard):
        """one shard of the exchange as measured (mfm_group_exchange_detail): dict"""
        d = ExchangeDetail()
        self._chk(self.lib.mfm_group_exchange_detail(self.h, shard, C.byref(d)), "mfm_group_exchange_detail")
        x = d.exchange_ms / d.timed_exchanges if d.timed_exchanges else None
        k = d.kernel_ms / d.timed_launches if d.timed_launches else None
        return {"device": d.device, "pci": d.pci_bus_id.decode(errors="replace"), "rccl_ranks": d.rccl_ranks,
                "exchange_ms": x, "kernel_ms": k, "timed_exchanges": d.timed_exchanges, "timed_launches": d.timed_launches,
                "bound": {0: None, 1: "kernel", 2: "exchange"}[d.bound]}


class Resampler:
    """mfm_resampler: rational resampler (+ optional DC blocker) for all channels of a PCM block."""

    def __init__(self, nr_channels, coeffs_q14, interpolate, decimate, max_in_samples, device=0, invert=False,
                 dc_pole=None, force_dot2=False):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = ResamplerConfig(MFM_ABI_VERSION, device, nr_channels, interpolate, decimate, max_in_samples,
                              int(invert), int(dc_pole is not None), float(dc_pole or 0.0),
                              MFM_RS_FORCE_DOT2 if force_dot2 else 0, 0)
        co = np.ascontiguousarray(coeffs_q14, dtype=np.int16)
        rc = self.lib.mfm_resampler_create(C.byref(self.h), C.byref(cfg), _i16p(co), co.size)
        if rc < 0:
            raise MfmError(rc, "mfm_resampler_create", self.lib.mfm_strerror(rc).decode())
        self.nr_channels = nr_channels

    def close(self):
        if self.h:
            self.lib.mfm_resampler_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_out(self):
        return self.lib.mfm_resampler_max_out(self.h)

    def process_host(self, pcm):
        """pcm: int16 [C][n] -> int16 [C][m]"""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.nr_channels, -1)
        cap = self.max_out()
        out = np.zeros((self.nr_channels, cap), np.int16)
        n = C.c_size_t()
        rc = self.lib.mfm_resampler_process_host(self.h, _i16p(a), a.shape[1], a.shape[1], _i16p(out), cap, C.byref(n))
        if rc < 0:
            raise MfmError(rc, "mfm_resampler_process_host", self.lib.mfm_strerror(rc).decode())
        return out[:, :n.value].copy()

    def process_device(self, d_pcm, in_stride, nr_in, stream=None):
        p, st, n = C.c_void_p(), C.c_size_t(), C.c_size_t()
        rc = self.lib.mfm_resampler_process_device(self.h, C.c_void_p(d_pcm), in_stride, nr_in, C.c_void_p(stream or 0),
                                                   C.byref(p), C.byref(st), C.byref(n))
        if rc < 0:
            raise MfmError(rc, "mfm_resampler_process_device", self.lib.mfm_strerror(rc).decode())
        return p.value, st.value, n.value


class F32Engine:
    """mfm_f32_*: the channel path on float32 IQ (FIR, derotation, discriminator in fp32)."""

    def __init__(self, sample_rate_hz, decimation, max_block_samples, device=0, want_iq=False, packed_fma=False,
                 tile_kernel=False):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = F32Config(MFM_ABI_VERSION, device, sample_rate_hz, decimation, max_block_samples,
                        (MFM_F32_WANT_IQ if want_iq else 0) | (MFM_F32_PACKED_FMA if packed_fma else 0) |
                        (MFM_F32_TILE_KERNEL if tile_kernel else 0))
        rc = self.lib.mfm_f32_create(C.byref(self.h), C.byref(cfg))
        if rc < 0:
            raise MfmError(rc, "mfm_f32_create", self.lib.mfm_strerror(rc).decode())
        self.want_iq = want_iq
        self.nr_channels = 0

    def _chk(self, rc, what):
        if rc < 0:
            raise MfmError(rc, what, self.lib.mfm_last_error().decode() or self.lib.mfm_strerror(rc).decode())
        return rc

    def close(self):
        if self.h:
            self.lib.mfm_f32_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_channel(self, offset_hz, lpf_taps, gain=1.0):
        t = np.ascontiguousarray(lpf_taps, dtype=np.float64)
        idx = self._chk(self.lib.mfm_f32_add_channel(self.h, int(offset_hz), t.ctypes.data_as(C.POINTER(C.c_double)),
                                                     t.size, float(gain)), "mfm_f32_add_channel")
        self.nr_channels = idx + 1
        return idx

    def commit(self):
        self._chk(self.lib.mfm_f32_commit(self.h), "mfm_f32_commit")

    def max_out(self):
        return self.lib.mfm_f32_max_out(self.h)

    def process_host(self, iq):
        """iq: float32 [n][2] (or flat interleaved) -> (pcm_f32 [C][m], pcm_i16 [C][m], iq_f32 [C][m][2] or None)"""
        a = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        n_in = a.size // 2
        cap = self.max_out()
        Cn = self.nr_channels
        pf = np.zeros((Cn, cap), np.float32)
        pi = np.zeros((Cn, cap), np.int16)
        qf = np.zeros((Cn, cap, 2), np.float32) if self.want_iq else None
        n = C.c_size_t()
        f32p = C.POINTER(C.c_float)
        self._chk(self.lib.mfm_f32_process_host(self.h, a.ctypes.data_as(f32p), n_in, pf.ctypes.data_as(f32p), _i16p(pi),
                                                qf.ctypes.data_as(f32p) if qf is not None else None, cap, C.byref(n)),
                  "mfm_f32_process_host")
        m = n.value
        return pf[:, :m].copy(), pi[:, :m].copy(), (qf[:, :m].copy() if qf is not None else None)

    def process_device(self, d_iq, nr_samples, stream=None):
        b = F32Block()
        self._chk(self.lib.mfm_f32_process_device(self.h, C.c_void_p(d_iq), nr_samples, C.c_void_p(stream or 0),
                                                  C.byref(b)), "mfm_f32_process_device")
        return b


class MuellerMuller:
    """mfm_mm_*: Mueller-Muller clock recovery (pager/mueller_muller.c) for all channels of a PCM block."""

    def __init__(self, nr_channels, kw, km, samples_per_bit, error_min, error_max, max_in_samples, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = MmConfig(MFM_ABI_VERSION, device, nr_channels, max_in_samples, kw, km, samples_per_bit, error_min,
                       error_max)
        rc = self.lib.mfm_mm_create(C.byref(self.h), C.byref(cfg))
        if rc < 0:
            raise MfmError(rc, "mfm_mm_create", self.lib.mfm_strerror(rc).decode())
        self.nr_channels = nr_channels

    def close(self):
        if self.h:
            self.lib.mfm_mm_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_host(self, pcm, nr_in):
        """pcm: int16 [C][stride] with stride > nr_in when the look-ahead sample is there -> list of per-channel
        decision arrays"""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.nr_channels, -1)
        cap = self.lib.mfm_mm_max_decisions(self.h)
        dec = np.zeros((self.nr_channels, cap), np.int16)
        cnt = np.zeros(self.nr_channels, np.uint32)
        rc = self.lib.mfm_mm_process_host(self.h, _i16p(a), a.shape[1], nr_in, _i16p(dec), cap,
                                          cnt.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc < 0:
            raise MfmError(rc, "mfm_mm_process_host", self.lib.mfm_strerror(rc).decode())
        return [dec[c, :cnt[c]].copy() for c in range(self.nr_channels)]

    def process_device(self, d_pcm, in_stride, nr_in, stream=None):
        """resident PCM -> (device pointer of the decisions, their row stride, device pointer of the counts)"""
        d_dec, stride, d_cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
        rc = self.lib.mfm_mm_process_device(self.h, C.c_void_p(d_pcm), in_stride, nr_in, C.c_void_p(stream or 0),
                                            C.byref(d_dec), C.byref(stride), C.byref(d_cnt))
        if rc < 0:
            raise MfmError(rc, "mfm_mm_process_device", self.lib.mfm_strerror(rc).decode())
        return d_dec.value, stride.value, d_cnt.value


class Pocsag:
    """mfm_pocsag: POCSAG slicer / sync / batch collection + BCH(31,21) for all channels of a 38 400 Hz PCM block."""

    def __init__(self, nr_channels, max_in_samples, device=0, max_events=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = PocsagConfig(MFM_ABI_VERSION, device, nr_channels, max_in_samples, max_events, 0)
        rc = self.lib.mfm_pocsag_create(C.byref(self.h), C.byref(cfg))
        if rc < 0:
            raise MfmError(rc, "mfm_pocsag_create", self.lib.mfm_strerror(rc).decode())
        self.nr_channels = nr_channels
        self.max_events = max_events or (max_in_samples // 2048 + 16)

    def close(self):
        if self.h:
            self.lib.mfm_pocsag_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_host(self, pcm):
        """pcm: int16 [C][n]; returns the events of this call as a structured array (POCSAG_EVENT_DTYPE)"""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.nr_channels, -1)
        rc = self.lib.mfm_pocsag_process_host(self.h, _i16p(a), a.shape[1], a.shape[1])
        if rc < 0:
            raise MfmError(rc, "mfm_pocsag_process_host", self.lib.mfm_strerror(rc).decode())
        return self.fetch_events()

    def process_device(self, d_pcm, in_stride, nr_in, stream=None):
        rc = self.lib.mfm_pocsag_process_device(self.h, C.c_void_p(d_pcm), in_stride, nr_in, C.c_void_p(stream or 0))
        if rc < 0:
            raise MfmError(rc, "mfm_pocsag_process_device", self.lib.mfm_strerror(rc).decode())

    def fetch_events(self):
        cap = self.nr_channels * self.max_events
        out = np.zeros(cap, POCSAG_EVENT_DTYPE)
        n = C.c_size_t()
        rc = self.lib.mfm_pocsag_fetch_events(self.h, out.ctypes.data, cap, C.byref(n))
        if rc < 0:
            raise MfmError(rc, "mfm_pocsag_fetch_events", self.lib.mfm_strerror(rc).decode())
        return out[:n.value].copy()


class Flex:
    """mfm_flex: FLEX sync 1 / FIW / sync 2 / block de-interleave for all channels of a 16 000 Hz PCM block."""

    def __init__(self, nr_channels, max_in_samples, device=0, max_events=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        cfg = FlexConfig(MFM_ABI_VERSION, device, nr_channels, max_in_samples, max_events, 0)
        rc = self.lib.mfm_flex_create(C.byref(self.h), C.byref(cfg))
        if rc < 0:
            raise MfmError(rc, "mfm_flex_create", self.lib.mfm_strerror(rc).decode())
        self.nr_channels = nr_channels
        self.max_events = max_events or (max_in_samples // 1024 + 8)
        self.max_frames = min(self.max_events, max_in_samples // 28672 + 2)

    def close(self):
        if self.h:
            self.lib.mfm_flex_destroy(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_host(self, pcm):
        """pcm: int16 [C][n]; returns (events, frames) of this call (FLEX_EVENT_DTYPE, FLEX_FRAME_DTYPE)"""
        a = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.nr_channels, -1)
        rc = self.lib.mfm_flex_process_host(self.h, _i16p(a), a.shape[1], a.shape[1])
        if rc < 0:
            raise MfmError(rc, "mfm_flex_process_host", self.lib.mfm_strerror(rc).decode())
        return self.fetch_events()

    def process_device(self, d_pcm, in_stride, nr_in, stream=None):
        rc = self.lib.mfm_flex_process_device(self.h, C.c_void_p(d_pcm), in_stride, nr_in, C.c_void_p(stream or 0))
        if rc < 0:
            raise MfmError(rc, "mfm_flex_process_device", self.lib.mfm_strerror(rc).decode())

    def fetch_events(self):
        cap_e, cap_f = self.nr_channels * self.max_events, self.nr_channels * self.max_frames
        ev = np.zeros(cap_e, FLEX_EVENT_DTYPE)
        fw = np.zeros(cap_f, FLEX_FRAME_DTYPE)
        ne, nf = C.c_size_t(), C.c_size_t()
        rc = self.lib.mfm_flex_fetch_events(self.h, ev.ctypes.data, cap_e, C.byref(ne), fw.ctypes.data, cap_f, C.byref(nf))
        if rc < 0:
            raise MfmError(rc, "mfm_flex_fetch_events", self.lib.mfm_strerror(rc).decode())
        return ev[:ne.value].copy(), fw[:nf.value].copy()


def bch3121_decode(words, device=0):
    """bch_code_decode on the GPU: returns (corrected uint32 array, rc uint8 array)"""
    lib = load_library()
    w = np.ascontiguousarray(words, dtype=np.uint32).copy()
    rc = np.zeros(w.size, np.uint8)
    r = lib.mfm_bch3121_decode_host(w.ctypes.data_as(C.POINTER(C.c_uint32)), rc.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    w.size, device)
    if r < 0:
        raise MfmError(r, "mfm_bch3121_decode_host", lib.mfm_strerror(r).decode())
    return w, rc


def hosttwin_bch3121_decode(word):
    lib = load_library()
    v = C.c_uint32(int(word))
    rc = lib.mfm_hosttwin_bch3121_decode(C.byref(v))
    return rc, v.value
