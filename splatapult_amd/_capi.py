"""ctypes binding of libmsplat.so (include/msplat.h).  No CPU fallback: if the library is
missing or no HIP device is present the product path raises, loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSPLAT_LIB_PATH: load another build of the same ABI (A/B timing of two commits on one GPU box); default: the in-tree build
LIB_PATH = os.environ.get("MSPLAT_LIB_PATH") or os.path.join(_HERE, "lib", "libmsplat.so")

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NO_CLOUD, ERR_NO_SORT, ERR_UNSUPPORTED, ERR_PAIR_OVERFLOW, ERR_IO = \
    -1, -2, -3, -4, -5, -6, -7, -8
ERR_PAIR_OVERFLOW_EARLIER = -9      # the call did its work; an EARLIER device-output frame had overflowed the pair buffer
FB_RGBA32F, FB_RGBA16F = 0, 1
ROP_NONE, ROP_RGBA8, ROP_RGBA16F = 0, 1, 2
RANK_AUTO, RANK_BALLOT = 0, 1
FRAMES_AUTO, FRAMES_SERIAL, FRAMES_IN_FLIGHT = 0, 1, 2
SPATIAL_AUTO, SPATIAL_ON, SPATIAL_OFF = 0, 1, 2
TWO_PASS_AUTO, TWO_PASS_ON, TWO_PASS_OFF = 0, 1, 2          # msplat_config.two_pass
BANDS_CONTIGUOUS, BANDS_INTERLEAVED, BANDS_BLOCK_INTERLEAVED, BANDS_ROOT_WEIGHTED = 0, 1, 2, 3
EXCHANGE_WIRE_FP16 = 1              # msplat_band_exchange flags
CU_ALL, CU_EVEN, CU_ODD = 0, 1, 2    # msplat_config.cu_partition
# "weighted": contiguous bands, rank 0 (the gather's root) weighted block_rows PERCENT of another rank (msplat.h)
BAND_KINDS = {"contiguous": BANDS_CONTIGUOUS, "interleaved": BANDS_INTERLEAVED, "block": BANDS_BLOCK_INTERLEAVED,
              "weighted": BANDS_ROOT_WEIGHTED}


class MsplatError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("msplat error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("fb_format", C.c_int32),
                ("srgb", C.c_int32), ("t_epsilon", C.c_float), ("pair_capacity", C.c_uint64),
                ("stream", C.c_void_p), ("enable_timing", C.c_int32), ("compositor_waves", C.c_int32),
                ("rank_mode", C.c_int32), ("frame_mode", C.c_int32), ("spatial_order", C.c_int32), ("async_submit", C.c_int32),
                ("two_pass", C.c_int32), ("cu_partition", C.c_int32)]


class AttrOffsets(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("pos_with_alpha", "r_sh0", "g_sh0", "b_sh0", "cov3_col0", "cov3_col1", "cov3_col2",
                 "r_sh1", "r_sh2", "r_sh3", "g_sh1", "g_sh2", "g_sh3", "b_sh1", "b_sh2", "b_sh3")]


class PlyLayout(C.Structure):
    _fields_ = [("vertex_size", C.c_uint32), ("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32),
                ("f_dc", C.c_int32 * 3), ("f_rest", C.c_int32 * 45), ("opacity", C.c_int32),
                ("scale", C.c_int32 * 3), ("rot", C.c_int32 * 4)]


class Stats(C.Structure):
    _fields_ = [("num_splats", C.c_uint64), ("sort_count", C.c_uint32), ("drawn", C.c_uint32),
                ("pairs", C.c_uint64), ("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("pair_capacity", C.c_uint64),
                ("device_bytes", C.c_uint64), ("pairs_tile16", C.c_uint64)]


class CompositeWork(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("work_items", "list_entries", "pair_words_fetched", "records_fetched", "records_composited",
                 "pixel_evals", "batches", "clocks_sum", "clocks_max", "inner_clocks_sum", "useful_evals")]


class Timings(C.Structure):
    _fields_ = [("sort_total", C.c_float), ("render_total", C.c_float), ("project", C.c_float),
                ("binning", C.c_float), ("composite", C.c_float), ("reserved", C.c_float * 3)]


# every symbol include/msplat.h and include/msplat_debug.h declare: (name, restype, argtypes)
_F16 = C.POINTER(C.c_float)
_U32P = C.POINTER(C.c_uint32)
SYMBOLS = [
    ("msplat_create", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Config)]),
    ("msplat_destroy", None, [C.c_void_p]),
    ("msplat_last_error", C.c_char_p, [C.c_void_p]),
    ("msplat_version_string", C.c_char_p, []),
    ("msplat_tile_size", C.c_int, []),
    ("msplat_upload_cloud", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(AttrOffsets), C.c_int]),
    ("msplat_upload_ply_vertices", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(PlyLayout), C.c_int]),
    ("msplat_upload_ply", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("msplat_download_cloud", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("msplat_set_band", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("msplat_set_band_cull", C.c_int, [C.c_void_p, C.c_int]),
    ("msplat_set_band_layout", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("msplat_band_plan", C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_int32)] * 4),
    ("msplat_band_plan_weighted", C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    ("msplat_band_root_weight", C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    ("msplat_get_stream", C.c_void_p, [C.c_void_p]),
    ("msplat_get_fb_format", C.c_int, [C.c_void_p]),
    ("msplat_debug_cu_partition", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_group_create", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(Config)]),
    ("msplat_group_destroy", None, [C.c_void_p]),
    ("msplat_group_last_error", C.c_char_p, [C.c_void_p]),
    ("msplat_group_size", C.c_uint32, [C.c_void_p]),
    ("msplat_group_context", C.c_void_p, [C.c_void_p, C.c_uint32]),
    ("msplat_group_peer_store", C.c_int, [C.c_void_p, C.c_uint32]),
    ("msplat_group_upload_cloud", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(AttrOffsets), C.c_int]),
    ("msplat_group_upload_gaussian_cloud", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_group_upload_ply", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("msplat_group_set_layout", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("msplat_group_set_band_cull", C.c_int, [C.c_void_p, C.c_int]),
    ("msplat_group_sort", C.c_int, [C.c_void_p, _F16, _F16, _F16, _F16]),
    ("msplat_group_render", C.c_int, [C.c_void_p, _F16, _F16, _F16, _F16, C.c_void_p, C.c_uint64, C.c_int]),
    ("msplat_group_synchronize", C.c_int, [C.c_void_p]),
    ("msplat_group_set_exchange", C.c_int, [C.c_void_p, C.c_int32]),
    ("msplat_group_get_exchange", C.c_int, [C.c_void_p]),
    ("msplat_band_exchange", C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32]),
    ("msplat_debug_band_exchange_loopback", C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_uint64,
                                                                                                C.c_int32, C.c_int32, C.c_int32]),
    ("msplat_sort", C.c_int, [C.c_void_p, _F16, _F16, _F16, _F16]),
    ("msplat_render", C.c_int, [C.c_void_p, _F16, _F16, _F16, _F16, C.c_void_p, C.c_uint64, C.c_int]),
    ("msplat_render_stereo", C.c_int, [C.c_void_p, _F16, _F16, _F16, _F16, _F16, _F16, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    ("msplat_synchronize", C.c_int, [C.c_void_p]),
    ("msplat_read_image", C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("msplat_points_create", C.c_void_p, [C.c_int]),
    ("msplat_points_destroy", None, [C.c_void_p]),
    ("msplat_points_import_ply", C.c_int, [C.c_void_p, C.c_char_p]),
    ("msplat_points_export_ply", C.c_int, [C.c_void_p, C.c_char_p]),
    ("msplat_points_init_debug", None, [C.c_void_p]),
    ("msplat_points_num", C.c_uint64, [C.c_void_p]),
    ("msplat_points_stride", C.c_uint32, [C.c_void_p]),
    ("msplat_points_data", C.c_void_p, [C.c_void_p]),
    ("msplat_upload_points", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("msplat_upload_point_cloud", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_set_point_sprite", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    ("msplat_set_depth_test", C.c_int, [C.c_void_p, C.c_int]),
    ("msplat_set_target_emulation", C.c_int, [C.c_void_p, C.c_int]),
    ("msplat_attach_cloud", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_stream_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_wait_event", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_sort_count", C.c_int, [C.c_void_p, _U32P]),
    ("msplat_get_sorted_indices", C.c_int, [C.c_void_p, _U32P, C.c_uint32]),
    ("msplat_get_sorted_keys", C.c_int, [C.c_void_p, _U32P, C.c_uint32]),
    ("msplat_get_storage_order", C.c_int, [C.c_void_p, _U32P, C.c_uint64, C.POINTER(C.c_int)]),
    ("msplat_debug_get_cull_boxes", C.c_int, [C.c_void_p, _U32P, _U32P, C.POINTER(C.c_int)]),
    ("msplat_get_stats", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("msplat_get_timings", C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    ("msplat_debug_get_projected", C.c_int, [C.c_void_p, _F16, _U32P, C.c_uint32]),
    ("msplat_debug_get_tile_lists", C.c_int, [C.c_void_p, _U32P, C.c_uint32, _U32P, C.c_uint64]),
    ("msplat_debug_get_tile_probe8", C.c_int, [C.c_void_p, _U32P, C.c_uint32]),
    ("msplat_set_tile_probe", C.c_int, [C.c_void_p, C.c_int]),
    ("msplat_debug_verify_order", C.c_int, [C.c_void_p, _U32P, _U32P]),
    ("msplat_debug_two_pass", C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
    ("msplat_get_two_pass_info", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("msplat_get_composite_work", C.c_int, [C.c_void_p, C.POINTER(CompositeWork)]),
    ("msplat_cloud_create", C.c_void_p, [C.c_int]),
    ("msplat_cloud_destroy", None, [C.c_void_p]),
    ("msplat_cloud_import_ply", C.c_int, [C.c_void_p, C.c_char_p]),
    ("msplat_cloud_export_ply", C.c_int, [C.c_void_p, C.c_char_p]),
    ("msplat_cloud_init_debug", C.c_int, [C.c_void_p]),
    ("msplat_cloud_prune", C.c_int, [C.c_void_p, _F16, C.c_uint32]),
    ("msplat_cloud_from_attributes", C.c_int, [C.c_void_p, C.c_uint64, _F16, _F16, _F16, _F16, _F16, _F16]),
    ("msplat_cloud_num_gaussians", C.c_uint64, [C.c_void_p]),
    ("msplat_cloud_stride", C.c_uint64, [C.c_void_p]),
    ("msplat_cloud_total_size", C.c_uint64, [C.c_void_p]),
    ("msplat_cloud_raw_data", C.c_void_p, [C.c_void_p]),
    ("msplat_cloud_has_full_sh", C.c_int, [C.c_void_p]),
    ("msplat_cloud_attr_offsets", C.c_int, [C.c_void_p, C.POINTER(AttrOffsets)]),
    ("msplat_upload_gaussian_cloud", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msplat_cameras_import_json", C.c_int, [C.c_char_p, _F16, _F16, C.c_uint32, _U32P]),
    ("msplat_cameras_floor_plane", C.c_int, [C.c_char_p, _F16, _F16]),
    ("msplat_vrconfig_import_json", C.c_int, [C.c_char_p, _F16]),
    ("msplat_vrconfig_export_json", C.c_int, [C.c_char_p, _F16]),
    ("msplat_find_config_file", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]),
    ("msplat_write_image", C.c_int, [C.c_char_p, _F16, C.c_int, C.c_int, C.c_int]),
    ("msplat_mat4_inverse", None, [_F16, _F16]),
    ("msplat_mat4_mul", None, [_F16, _F16, _F16]),
    ("msplat_perspective", None, [C.c_float, C.c_float, C.c_float, C.c_float, _F16]),
    ("msplat_create_projection", None, [C.c_float] * 6 + [_F16]),
]

_LIB = None


def lib():
    """Load libmsplat.so.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            try:
                fn = getattr(L, name)
            except AttributeError:
                # another build of the ABI loaded for an A/B (MSPLAT_LIB_PATH: e.g. an earlier round's tree) may predate an entry
                # point -- it then fails at its first use; the in-tree library must export everything
                if os.environ.get("MSPLAT_LIB_PATH"):
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class EarlierFrameOverflow(UserWarning):
    """msplat_sort / msplat_render succeeded, but an earlier device-output render on the context had overflowed the pair
    buffer (MSPLAT_ERR_PAIR_OVERFLOW_EARLIER): that frame lacks splats and should be rendered again"""


def check(ctx, rc):
    if rc == ERR_PAIR_OVERFLOW_EARLIER:          # a warning about a past frame: this call's result is valid
        import warnings
        msg = lib().msplat_last_error(ctx)
        warnings.warn(EarlierFrameOverflow(msg.decode() if msg else "an earlier frame overflowed the pair buffer"), stacklevel=3)
        return
    if rc != OK:
        msg = lib().msplat_last_error(ctx)
        raise MsplatError(rc, msg.decode() if msg else "")


def band_plan(kind, rows_full, world, rank, block_rows=1):
    """(first_row, row_count, block, stride) of rank `rank` of `world` over rows_full bin rows; kind = "contiguous" |
    "interleaved" | "block" (blocks of block_rows rows dealt round-robin).  Host arithmetic (msplat_band_plan)."""
    out = [C.c_int32() for _ in range(4)]
    rc = lib().msplat_band_plan(BAND_KINDS[kind] if isinstance(kind, str) else int(kind), rows_full, world, rank, block_rows,
                                *[C.byref(o) for o in out])
    check(None, rc)
    return tuple(o.value for o in out)


def band_plan_weighted(rows_full, weights):
    """bounds[world + 1] of contiguous bands with row counts proportional to `weights` (msplat_band_plan_weighted)"""
    w = (C.c_float * len(weights))(*[float(x) for x in weights])
    out = (C.c_int32 * (len(weights) + 1))()
    check(None, lib().msplat_band_plan_weighted(rows_full, len(weights), w, out))
    return list(out)


def band_root_weight(rows_full, world, fixed_ms, ms_per_row, row_bytes, link_gbps=153.0, overlap=True):
    """root weight (percent) for the "weighted" layout from the linear cost model of msplat_band_root_weight"""
    return int(lib().msplat_band_root_weight(rows_full, world, float(fixed_ms), float(ms_per_row), float(row_bytes), float(link_gbps),
                                             1 if overlap else 0))


def band_rows(first, count, block, stride, rows_full=None):
    """the bin rows a layout owns, ascending (what msplat_set_band_layout's parameters mean)"""
    rows, v = [], 0
    while True:
        t = first + (v // block) * stride + (v % block)
        if (count and v >= count) or (rows_full is not None and t >= rows_full) or (not count and rows_full is None):
            break
        rows.append(t)
        v += 1
    return rows
