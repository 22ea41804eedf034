"""Python host binding of libdmz_hip.so (the C-ABI of include/dmz_hip.h).

The directory name `card.io-dmz_amd` is not a Python identifier; load it with
`__graft_entry__.load_package()` (importlib by path, module name `dmz_amd`).

This module is plumbing only: numpy/torch buffers in, ctypes calls to the HIP
library out.  There is NO CPU fallback here -- if the shared library or a GPU is
missing every entry point raises.  (The CPU oracle lives in oracle/ and is never
imported from this package.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# DMZ_HIP_LIB: developer override used by tools/ablate.sh to time kernel variants
LIB_PATH = os.environ.get("DMZ_HIP_LIB") or os.path.join(HERE, "libdmz_hip.so")

FRAME_W, FRAME_H = 640, 480
CARD_W, CARD_H = 428, 270
CARD_BYTES = CARD_W * CARD_H
FRAME_BYTES = FRAME_W * FRAME_H

ORIENTATION_PORTRAIT = 1
ORIENTATION_PORTRAIT_UPSIDE_DOWN = 2
ORIENTATION_LANDSCAPE_RIGHT = 3
ORIENTATION_LANDSCAPE_LEFT = 4

FLAG_USABLE, FLAG_UPSIDE_DOWN, FLAG_VSEG_OK, FLAG_WARPED = 1, 2, 4, 8
SCAN_ONLY_WARPED, SCAN_SKIP_NUMBER = 1, 2
EXPIRY_CONV_F32, EXPIRY_CONV_BF16X3, EXPIRY_CONV_BF16, EXPIRY_CONV_F16X3 = 0, 1, 2, 3
OPT_TRUNCATE_CORNERS = 1
OPT_UPSAMPLE = 2
OPT_EIGEN_SSE2 = 4
OPT_EIGEN_SCALAR = 8  # per-call override of set_reference_flavour(1)
FLAG_FAULT = 16       # a device self-check did not settle (include/dmz_hip.h)
STAGES = ("detect", "geometry", "warp", "vseg", "hseg", "digits", "expiry_seg", "expiry_cat")

# mirror of struct dmz_hip_frame_result (include/dmz_hip.h), 1024 bytes
RESULT_DTYPE = np.dtype([
    ("found", "<i4", (4,)), ("rho", "<f4", (4,)), ("theta", "<f4", (4,)),
    ("corners", "<f4", (8,)), ("found_all", "<i4"), ("flags", "<i4"),
    ("vseg_score", "<f4"), ("vseg_y_offset", "<i4"), ("pattern_type", "<i4"),
    ("n_offsets", "<i4"), ("offsets", "<u2", (16,)), ("hseg_score", "<f4"),
    ("number_width", "<f4"), ("pattern_offset", "<i4"), ("number_score", "<f4"),
    ("digits", "u1", (16,)), ("scores", "<f4", (16, 10)),
    ("expiry_month", "<i4"), ("expiry_year", "<i4"), ("reserved", "u1", (208,)),
])
assert RESULT_DTYPE.itemsize == 1024

# mirror of struct dmz_hip_expiry_group / dmz_hip_expiry_result (include/dmz_hip.h)
EXPIRY_MAX_GROUPS = 8
EXPIRY_GROUP_DTYPE = np.dtype([
    ("top", "<i2"), ("left", "<i2"), ("width", "<i2"), ("height", "<i2"),
    ("char_top", "<i2", (5,)), ("char_left", "<i2", (5,)),
    ("stripe_base_row", "<i2"), ("reserved", "<i2"), ("scores", "<f4", (4, 10)),
])
assert EXPIRY_GROUP_DTYPE.itemsize == 192
EXPIRY_DTYPE = np.dtype([
    ("n_groups", "<i4"), ("n_found", "<i4"), ("n_stripes", "<i4"),
    ("stripe_base_row", "<i4", (3,)), ("stripe_sum", "<i8", (3,)),
    ("categorised", "<i4"), ("reserved", "<i4"),
    ("groups", EXPIRY_GROUP_DTYPE, (EXPIRY_MAX_GROUPS,)),
])
assert EXPIRY_DTYPE.itemsize == 1592

# mirror of struct dmz_hip_session_result (include/dmz_hip.h)
SESSION_DTYPE = np.dtype([
    ("complete", "<i4"), ("complete_frame", "<i4"), ("number_frame", "<i4"), ("n_numbers", "<i4"),
    ("predictions", "u1", (16,)), ("card_type", "<i4"), ("expiry_month", "<i4"), ("expiry_year", "<i4"),
    ("count15", "<i4"), ("count16", "<i4"), ("usable_frames", "<i4"), ("n_expiry_groups", "<i4"),
    ("vseg_y_offset", "<i4"), ("n_offsets", "<i4"), ("offsets", "<u2", (16,)), ("number_width", "<f4"), ("reserved", "<i4", (6,)),
])
assert SESSION_DTYPE.itemsize == 128

# every symbol include/dmz_hip.h declares
EXPORTS = (
    "dmz_hip_device_count", "dmz_hip_context_create", "dmz_hip_context_destroy",
    "dmz_hip_synchronize", "dmz_hip_set_stream", "dmz_hip_last_error",
    "dmz_hip_detect_batch", "dmz_hip_transform_batch", "dmz_hip_scan_cards_batch",
    "dmz_hip_pipeline_batch", "dmz_hip_calc_persp_transform", "dmz_hip_warp_perspective_batch",
    "dmz_hip_apply_vseg_model", "dmz_hip_apply_digit_model", "dmz_hip_synth_frames",
    "dmz_hip_synth_cards", "dmz_hip_set_profiling", "dmz_hip_get_stage_times",
    "dmz_hip_malloc", "dmz_hip_free", "dmz_hip_memcpy_h2d", "dmz_hip_memcpy_d2h",
    "dmz_hip_scan_expiry_batch", "dmz_hip_pipeline_expiry_batch",
    "dmz_hip_apply_slash_model", "dmz_hip_apply_expiry_model", "dmz_hip_scan_sessions_batch",
    "dmz_hip_deinterleave_c2", "dmz_hip_deinterleave_rgba_to_r", "dmz_hip_ycbcr_to_rgb",
    "dmz_hip_scores_batch", "dmz_hip_blur_cards_batch", "dmz_hip_set_expiry_conv", "dmz_hip_set_two_queues",
    "dmz_hip_shard_range", "dmz_hip_comm_unique_id", "dmz_hip_comm_init", "dmz_hip_comm_destroy",
    "dmz_hip_gather_records", "dmz_hip_gather_wait", "dmz_hip_expiry_sort_positions",
    "dmz_hip_categorize_expiry_groups_batch", "dmz_hip_scharr3_dx_abs", "dmz_hip_best_n_hseg_batch",
    "dmz_hip_set_reference_flavour",
)
# test / developer entry points (include/dmz_hip_test.h): not part of the drop-in boundary
TEST_EXPORTS = ("dmz_hip_debug_fill_lds",)


class DmzHipError(RuntimeError):
    pass


def shard_range(n_total, world, rank):
    """[first, first + count) of `rank` (dmz_hip_shard_range: the C-ABI's split, no device needed)"""
    first, count = C.c_int64(), C.c_int64()
    load_library().dmz_hip_shard_range(n_total, world, rank, C.byref(first), C.byref(count))
    return first.value, count.value


def comm_unique_id():
    """the 128-byte RCCL id rank 0 hands to every rank's Context.comm_init"""
    buf = (C.c_char * 128)()
    rc = load_library().dmz_hip_comm_unique_id(buf)
    if rc != 0:
        raise DmzHipError("dmz_hip_comm_unique_id failed with %d (librccl not loadable?)" % rc)
    return bytes(buf)


def build(force=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libdmz_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmzHipError("%s is missing: run `make -C %s` (the HIP extension is required; "
                          "there is no CPU fallback)" % (LIB_PATH, HERE))
    lib = C.CDLL(LIB_PATH)
    vp, i, sz, u64 = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64
    lib.dmz_hip_device_count.restype = i
    lib.dmz_hip_context_create.argtypes = [i, C.POINTER(vp)]
    lib.dmz_hip_context_destroy.argtypes = [vp]
    lib.dmz_hip_context_destroy.restype = None
    lib.dmz_hip_synchronize.argtypes = [vp]
    lib.dmz_hip_set_stream.argtypes = [vp, vp]
    lib.dmz_hip_set_expiry_conv.argtypes = [vp, i]
    lib.dmz_hip_set_two_queues.argtypes = [vp, i]
    lib.dmz_hip_last_error.argtypes = [vp]
    lib.dmz_hip_last_error.restype = C.c_char_p
    lib.dmz_hip_detect_batch.argtypes = [vp, vp, sz, i, i, i, vp, vp, sz, i, i, i, vp]
    lib.dmz_hip_transform_batch.argtypes = [vp, vp, sz, i, i, i, i, i, i, vp, vp, sz]
    lib.dmz_hip_scan_cards_batch.argtypes = [vp, vp, sz, i, i, vp]
    lib.dmz_hip_best_n_hseg_batch.argtypes = [vp, vp, sz, i, vp]
    lib.dmz_hip_pipeline_batch.argtypes = [vp, vp, sz, i, i, i, i, i, i, vp, sz, vp]
    lib.dmz_hip_scan_expiry_batch.argtypes = [vp, vp, sz, i, vp, vp]
    lib.dmz_hip_pipeline_expiry_batch.argtypes = [vp, vp, sz, i, i, i, i, i, i, vp, sz, vp, vp]
    lib.dmz_hip_blur_cards_batch.argtypes = [vp, vp, sz, i, i, vp, i]
    lib.dmz_hip_scores_batch.argtypes = [vp, vp, sz, i, i, i, i, i, vp, vp]
    lib.dmz_hip_deinterleave_c2.argtypes = [vp, vp, sz, vp, vp]
    lib.dmz_hip_deinterleave_rgba_to_r.argtypes = [vp, vp, vp, sz]
    lib.dmz_hip_ycbcr_to_rgb.argtypes = [vp, vp, vp, vp, sz, i, vp]
    lib.dmz_hip_scan_sessions_batch.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, vp]
    lib.dmz_hip_apply_slash_model.argtypes = [vp, vp, i, vp]
    lib.dmz_hip_apply_expiry_model.argtypes = [vp, vp, i, vp]
    lib.dmz_hip_expiry_sort_positions.argtypes = [vp, vp, vp, vp, i, i, i, vp, vp]
    lib.dmz_hip_categorize_expiry_groups_batch.argtypes = [vp, vp, sz, i, vp]
    lib.dmz_hip_scharr3_dx_abs.argtypes = [vp, vp, i, i, i, vp, i]
    lib.dmz_hip_calc_persp_transform.argtypes = [vp, vp, vp, vp]
    lib.dmz_hip_warp_perspective_batch.argtypes = [vp, vp, sz, i, i, i, i, vp, vp, sz]
    lib.dmz_hip_apply_vseg_model.argtypes = [vp, vp, i, vp]
    lib.dmz_hip_apply_digit_model.argtypes = [vp, i, vp, i, vp]
    lib.dmz_hip_synth_frames.argtypes = [vp, u64, u64, i, vp]
    lib.dmz_hip_synth_cards.argtypes = [vp, u64, u64, i, vp]
    lib.dmz_hip_debug_fill_lds.argtypes = [vp, C.c_uint32]
    lib.dmz_hip_set_reference_flavour.argtypes = [vp, i]
    lib.dmz_hip_set_profiling.argtypes = [vp, i]
    lib.dmz_hip_get_stage_times.argtypes = [vp, vp, vp, i]
    lib.dmz_hip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.dmz_hip_free.argtypes = [vp, vp]
    lib.dmz_hip_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.dmz_hip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.dmz_hip_shard_range.argtypes = [C.c_int64, i, i, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dmz_hip_shard_range.restype = None
    lib.dmz_hip_comm_unique_id.argtypes = [vp]
    lib.dmz_hip_comm_init.argtypes = [vp, vp, i, i]
    lib.dmz_hip_comm_destroy.argtypes = [vp]
    lib.dmz_hip_gather_records.argtypes = [vp, vp, sz, C.c_int64, i, vp, i]
    lib.dmz_hip_gather_wait.argtypes = [vp, i, i]
    _lib = lib
    return lib


def _ptr(a):
    """Address of a numpy array, a torch tensor, a raw int address or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):  # torch tensor (device or host)
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(type(a))


class DeviceBuffer:
    """Raw HBM allocation through the C-ABI (for hosts without torch)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._check(ctx.lib.dmz_hip_malloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.dmz_hip_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count=None):
        dtype = np.dtype(dtype)
        count = self.nbytes // dtype.itemsize if count is None else count
        out = np.empty(count, dtype)
        self.ctx._check(self.ctx.lib.dmz_hip_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.dmz_hip_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """dmz_hip_context: the MI355X twin of the reference's dmz_context / mz handle."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.dmz_hip_device_count() <= 0:
            raise DmzHipError("no HIP device visible: the MI355X path cannot run (no CPU fallback)")
        h = C.c_void_p()
        rc = self.lib.dmz_hip_context_create(device, C.byref(h))
        if rc != 0:
            raise DmzHipError("dmz_hip_context_create(%d) failed with %d" % (device, rc))
        self.h = h

    def close(self):
        if self.h:
            self.lib.dmz_hip_context_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise DmzHipError("dmz_hip error %d: %s" % (rc, self.lib.dmz_hip_last_error(self.h).decode()))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def synchronize(self):
        self._check(self.lib.dmz_hip_synchronize(self.h))

    # ---- frames sharded over the GPUs of a node (include/dmz_hip.h: one context per GPU) ----
    def comm_init(self, world=1, rank=0, unique_id=None):
        """RCCL communicator of this context; `unique_id` = the 128 bytes rank 0 got from comm_unique_id()"""
        buf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        self._check(self.lib.dmz_hip_comm_init(self.h, buf, world, rank))

    def comm_destroy(self):
        self._check(self.lib.dmz_hip_comm_destroy(self.h))

    def gather_records(self, local, record_bytes, n_total, root=0, root_dst=None, slot=0):
        """asynchronous gather of this rank's shard of n_total records on `root` (device pointers); `slot` names it for the wait"""
        self._check(self.lib.dmz_hip_gather_records(self.h, _ptr(local), record_bytes, n_total, root, _ptr(root_dst), slot))

    def gather_wait(self, slot=-1, host_sync=True):
        """wait for the slot's last gather (slot < 0: all of them)"""
        self._check(self.lib.dmz_hip_gather_wait(self.h, slot, int(host_sync)))

    def set_stream(self, stream_handle):
        self._check(self.lib.dmz_hip_set_stream(self.h, stream_handle))

    def set_expiry_conv(self, mode):
        """arithmetic of the expiry CNN's convolutions: EXPIRY_CONV_F16X3 (default) / _F32 / _BF16X3 / _BF16"""
        self._check(self.lib.dmz_hip_set_expiry_conv(self.h, mode))

    def set_two_queues(self, on):
        """pipeline_expiry: expiry segmentation on a second device queue beside hseg + digits (default on)"""
        self._check(self.lib.dmz_hip_set_two_queues(self.h, int(on)))

    def set_profiling(self, on):
        self._check(self.lib.dmz_hip_set_profiling(self.h, int(on)))

    def stage_times(self, reset=True):
        ms = np.zeros(len(STAGES), np.float32)
        cnt = np.zeros(len(STAGES), np.int32)
        self._check(self.lib.dmz_hip_get_stage_times(self.h, ms.ctypes.data, cnt.ctypes.data, int(reset)))
        return {s: (float(ms[k]), int(cnt[k])) for k, s in enumerate(STAGES)}

    # ---- batched stages (pointers: numpy host arrays, torch tensors, DeviceBuffer.ptr ints) ----
    def detect(self, y, n, results, width=FRAME_W, height=FRAME_H, orientation=ORIENTATION_LANDSCAPE_RIGHT,
               cb=None, cr=None, frame_stride=None, row_stride=None):
        row_stride = row_stride or width
        frame_stride = frame_stride or row_stride * height
        self._check(self.lib.dmz_hip_detect_batch(
            self.h, _ptr(y), frame_stride, row_stride, width, height, _ptr(cb), _ptr(cr),
            (width // 2) * (height // 2), width // 2, n, orientation, _ptr(results)))

    def transform(self, plane, n, results, cards, width=FRAME_W, height=FRAME_H,
                  orientation=ORIENTATION_LANDSCAPE_RIGHT, options=0, frame_stride=None, row_stride=None):
        row_stride = row_stride or width
        frame_stride = frame_stride or row_stride * height
        self._check(self.lib.dmz_hip_transform_batch(
            self.h, _ptr(plane), frame_stride, row_stride, width, height, n, orientation, options,
            _ptr(results), _ptr(cards), CARD_BYTES))

    def scan_cards(self, cards, n, results, only_warped=False, skip_number=False):
        mode = (SCAN_ONLY_WARPED if only_warped else 0) | (SCAN_SKIP_NUMBER if skip_number else 0)
        self._check(self.lib.dmz_hip_scan_cards_batch(self.h, _ptr(cards), CARD_BYTES, n, mode, _ptr(results)))

    def best_n_hseg(self, cards, n, results):
        """best_n_hseg (n_hseg.cpp:88) at each record's vseg_y_offset / pattern_type (records with FLAG_VSEG_OK)"""
        self._check(self.lib.dmz_hip_best_n_hseg_batch(self.h, _ptr(cards), CARD_BYTES, n, _ptr(results)))

    def pipeline(self, y, n, results, cards=None, width=FRAME_W, height=FRAME_H,
                 orientation=ORIENTATION_LANDSCAPE_RIGHT, options=0):
        self._check(self.lib.dmz_hip_pipeline_batch(
            self.h, _ptr(y), width * height, width, width, height, n, orientation, options,
            _ptr(cards), CARD_BYTES, _ptr(results)))

    def scan_expiry(self, cards, n, results, expiry):
        self._check(self.lib.dmz_hip_scan_expiry_batch(self.h, _ptr(cards), CARD_BYTES, n, _ptr(results),
                                                       _ptr(expiry)))

    def pipeline_expiry(self, y, n, results, expiry, cards=None, width=FRAME_W, height=FRAME_H,
                        orientation=ORIENTATION_LANDSCAPE_RIGHT, options=0):
        self._check(self.lib.dmz_hip_pipeline_expiry_batch(
            self.h, _ptr(y), width * height, width, width, height, n, orientation, options,
            _ptr(cards), CARD_BYTES, _ptr(results), _ptr(expiry)))

    def blur_cards(self, rgb, n, sessions, unblur_digits, channels=3):
        self._check(self.lib.dmz_hip_blur_cards_batch(self.h, _ptr(rgb), CARD_BYTES * channels, channels, n,
                                                      _ptr(sessions), unblur_digits))

    def scores(self, y, n, focus, brightness, width=FRAME_W, height=FRAME_H, use_full_image=False):
        self._check(self.lib.dmz_hip_scores_batch(self.h, _ptr(y), width * height, width, width, height, n,
                                                  int(use_full_image), _ptr(focus), _ptr(brightness)))

    def deinterleave_c2(self, interleaved, n_pairs, channel1, channel2):
        self._check(self.lib.dmz_hip_deinterleave_c2(self.h, _ptr(interleaved), n_pairs, _ptr(channel1), _ptr(channel2)))

    def deinterleave_rgba_to_r(self, source, dest, size):
        self._check(self.lib.dmz_hip_deinterleave_rgba_to_r(self.h, _ptr(source), _ptr(dest), size))

    def ycbcr_to_rgb(self, y, cb, cr, n_pixels, rgb, channels=3):
        self._check(self.lib.dmz_hip_ycbcr_to_rgb(self.h, _ptr(y), _ptr(cb), _ptr(cr), n_pixels, channels, _ptr(rgb)))

    def scan_sessions(self, results, expiry, n_sessions, frames_per_session, out, scan_expiry=True,
                      frame_interval_ms=33, now_year=None, now_month=None, allow_past_expiry=False):
        if now_year is None or now_month is None:  # the reference reads localtime (expiry_categorize.cpp:262-270)
            import datetime
            today = datetime.date.today()
            now_year = today.year if now_year is None else now_year
            now_month = today.month if now_month is None else now_month
        self._check(self.lib.dmz_hip_scan_sessions_batch(
            self.h, _ptr(results), _ptr(expiry), n_sessions, frames_per_session, int(scan_expiry),
            frame_interval_ms, now_year, now_month, int(allow_past_expiry), _ptr(out)))

    def calc_persp_transform(self, src_pts, dst_pts):
        s = np.ascontiguousarray(src_pts, np.float32).reshape(8)
        d = np.ascontiguousarray(dst_pts, np.float32).reshape(8)
        m = np.empty(9, np.float32)
        self._check(self.lib.dmz_hip_calc_persp_transform(self.h, s.ctypes.data, d.ctypes.data, m.ctypes.data))
        return m

    def warp_perspective(self, plane, n, matrices, cards, width=FRAME_W, height=FRAME_H):
        self._check(self.lib.dmz_hip_warp_perspective_batch(
            self.h, _ptr(plane), width * height, width, width, height, n, _ptr(matrices), _ptr(cards), CARD_BYTES))

    def apply_vseg_model(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 204)
        out = np.empty((x.shape[0], 3), np.float32)
        self._check(self.lib.dmz_hip_apply_vseg_model(self.h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def apply_digit_model(self, model, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 27 * 19)
        out = np.empty((x.shape[0], 10), np.float32)
        self._check(self.lib.dmz_hip_apply_digit_model(self.h, model, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def apply_slash_model(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 176)
        out = np.empty((x.shape[0], 2), np.float32)
        self._check(self.lib.dmz_hip_apply_slash_model(self.h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def apply_expiry_model(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 176)
        out = np.empty((x.shape[0], 10), np.float32)
        self._check(self.lib.dmz_hip_apply_expiry_model(self.h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def categorize_expiry_groups(self, cards, n, expiry):
        """scores of caller-supplied groups (host records, in / out): dmz_hip_categorize_expiry_groups_batch"""
        assert isinstance(expiry, np.ndarray)
        self._check(self.lib.dmz_hip_categorize_expiry_groups_batch(self.h, _ptr(cards), CARD_BYTES, n, expiry.ctypes.data))

    def scharr3_dx_abs(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.zeros(img.shape, np.int16)
        self._check(self.lib.dmz_hip_scharr3_dx_abs(self.h, img.ctypes.data, img.shape[1], img.shape[1], img.shape[0],
                                                    out.ctypes.data, img.shape[1]))
        return out

    def expiry_sort_positions(self, keys, lens, kind=0, marks=None):
        """(pos, flags) of dmz_hip_expiry_sort_positions for int32 key lists [n_lists, stride]."""
        keys = np.ascontiguousarray(keys, np.int32)
        if marks is not None:
            marks = np.ascontiguousarray(marks, np.int32)
            assert marks.shape == keys.shape
        lens = np.ascontiguousarray(lens, np.int32)
        pos = np.zeros(keys.shape, np.int32)
        flags = np.zeros(keys.shape[0], np.int32)
        self._check(self.lib.dmz_hip_expiry_sort_positions(self.h, keys.ctypes.data,
                                                           marks.ctypes.data if marks is not None else None,
                                                           lens.ctypes.data, keys.shape[0],
                                                           keys.shape[1], kind, pos.ctypes.data, flags.ctypes.data))
        return pos, flags

    def synth_frames(self, seed, first, n, y_dev):
        self._check(self.lib.dmz_hip_synth_frames(self.h, seed, first, n, _ptr(y_dev)))

    def set_reference_flavour(self, flavour):
        """0: Eigen's scalar order (default); 1: a stock x86-64 build's SSE2 order (DMZ_HIP_OPT_EIGEN_SSE2)"""
        self._check(self.lib.dmz_hip_set_reference_flavour(self.h, flavour))

    def debug_fill_lds(self, word=0xFFFFFFFF):
        self._check(self.lib.dmz_hip_debug_fill_lds(self.h, word))

    def synth_cards(self, seed, first, n, cards_dev):
        self._check(self.lib.dmz_hip_synth_cards(self.h, seed, first, n, _ptr(cards_dev)))
