"""ctypes mirror of include/sdm_c.h.  No compute happens here: every method forwards to the HIP
library and raises SdmError on a non-zero status.  There is no CPU fallback -- if
lib/libsdm_hip.so is missing this module fails loudly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_NEIGHBOURS = 64


class SdmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sdm error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("lambdaG", C.c_float), ("lambdaL", C.c_float), ("lambdaTheta", C.c_float),
                ("lambdaN", C.c_int), ("theta_var", C.c_double)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("W", C.c_int), ("H", C.c_int), ("max_keyframes", C.c_int),
                ("max_neighbours", C.c_int), ("batch_capacity", C.c_int), ("with_pointset", C.c_int),
                ("ext_depth_pool", C.c_void_p), ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("searches", C.c_longlong), ("candidates", C.c_longlong), ("gate_pass", C.c_longlong),
                ("hypotheses", C.c_longlong), ("fused", C.c_longlong), ("mask_waves", C.c_longlong),
                ("mask_steps", C.c_longlong), ("mask_row_mismatch", C.c_longlong), ("open_pixels", C.c_longlong),
                ("table_stagings", C.c_longlong)]


def lib_path():
    # SDM_LIB_PATH: A/B a differently-built engine (kernel experiments); still a HIP library
    return os.environ.get("SDM_LIB_PATH") or os.path.join(_HERE, "lib", "libsdm_hip.so")


_lib = None

# every symbol include/sdm_c.h declares: (name, restype, argtypes)
_f32p, _u8p, _ip = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int)
_ctx = C.c_void_p
SYMBOLS = [
    ("sdm_default_params", None, [C.POINTER(Params)]),
    ("sdm_default_config", None, [C.POINTER(Config)]),
    ("sdm_depth_pool_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("sdm_create", C.c_int, [C.POINTER(_ctx), C.POINTER(Config)]),
    ("sdm_destroy", None, [_ctx]),
    ("sdm_last_error", C.c_char_p, []),
    ("sdm_set_params", C.c_int, [_ctx, C.POINTER(Params)]),
    ("sdm_set_stream", C.c_int, [_ctx, C.c_void_p]),
    ("sdm_synchronize", C.c_int, [_ctx]),
    ("sdm_device_count", C.c_int, []),
    ("sdm_upload_keyframe", C.c_int, [_ctx, C.c_int, _u8p, _f32p, _f32p, C.c_float, _f32p, _f32p]),
    ("sdm_upload_image", C.c_int, [_ctx, C.c_int, _u8p, _f32p, _f32p]),
    ("sdm_upload_image_rgb", C.c_int, [_ctx, C.c_int, _u8p, C.c_int, _f32p, _f32p, _f32p]),
    ("sdm_upload_image_device", C.c_int, [_ctx, C.c_int, C.c_void_p, _f32p, _f32p]),
    ("sdm_upload_images_batch", C.c_int, [_ctx, C.c_int, _ip, C.POINTER(C.c_void_p), _f32p, _f32p]),
    ("sdm_upload_images_rgb_batch", C.c_int, [_ctx, C.c_int, _ip, C.POINTER(C.c_void_p), C.c_int, _f32p, _f32p, _f32p]),
    ("sdm_host_alloc", C.c_void_p, [C.c_size_t]),
    ("sdm_host_free", None, [C.c_void_p]),
    ("sdm_set_pose", C.c_int, [_ctx, C.c_int, _f32p]),
    ("sdm_download_inputs", C.c_int, [_ctx, C.c_int, _u8p, _f32p, _f32p, _f32p]),
    ("sdm_search_fuse", C.c_int, [_ctx, C.c_int, _ip, C.c_int, _ip, _f32p, _f32p, _f32p]),
    ("sdm_intra_check", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_intra_grow", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_recon", C.c_int, [_ctx, C.c_int, _ip, C.c_int, _ip, _f32p, _f32p, _f32p]),
    ("sdm_inter_check", C.c_int, [_ctx, C.c_int, _ip, C.c_int, _ip, C.c_int]),
    ("sdm_pointset", C.c_int, [_ctx, C.c_int, _ip, C.c_int]),
    ("sdm_inter_check_pointset", C.c_int, [_ctx, C.c_int, _ip, C.c_int, _ip, C.c_int]),
    ("sdm_upload_depth", C.c_int, [_ctx, C.c_int, _f32p, _f32p]),
    ("sdm_download_depth", C.c_int, [_ctx, C.c_int, _f32p, _f32p]),
    ("sdm_download_checked", C.c_int, [_ctx, C.c_int, _f32p]),
    ("sdm_download_pointset", C.c_int, [_ctx, C.c_int, _f32p]),
    ("sdm_depth_pool_ptr", C.c_void_p, [_ctx]),
    ("sdm_assume_pipeline_maps", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_mark_depth_present", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_comm_unique_id", C.c_int, [C.POINTER(C.c_ubyte)]),
    ("sdm_comm_init", C.c_int, [_ctx, C.POINTER(C.c_ubyte), C.c_int, C.c_int]),
    ("sdm_comm_attach", C.c_int, [_ctx, C.c_void_p]),
    ("sdm_comm_destroy", C.c_int, [_ctx]),
    ("sdm_comm_info", C.c_int, [_ctx, _ip, _ip]),
    ("sdm_exchange_halo_begin", C.c_int, [_ctx, C.c_int, _ip, _ip, C.c_int, _ip, _ip]),
    ("sdm_exchange_wait", C.c_int, [_ctx]),
    ("sdm_exchange_halo", C.c_int, [_ctx, C.c_int, _ip, _ip, C.c_int, _ip, _ip]),
    ("sdm_allgather_depth", C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _ip, _ip]),
    ("sdm_allgather_begin", C.c_int, [_ctx, C.c_int]),
    ("sdm_allgather_piece", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_allgather_finish", C.c_int, [_ctx, C.c_int, _ip, _ip]),
    ("sdm_comm_all_ok", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_comm_all_max", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_exchange_compact", C.c_int, [_ctx, C.c_int]),
    ("sdm_exchange_mismatches", C.c_int, [_ctx, _ip]),
    ("sdm_compact_sources_ready", C.c_int, [_ctx, C.c_int, _ip, _ip]),
    ("sdm_compact_pack_host", C.c_int, [_ctx, C.c_int, _f32p]),
    ("sdm_compact_unpack_host", C.c_int, [_ctx, C.c_int, _f32p, _ip]),
    ("sdm_active_count", C.c_int, [_ctx, C.c_int, _ip]),
    ("sdm_download_active_list", C.c_int, [_ctx, C.c_int, C.POINTER(C.c_uint), C.c_int, _ip, C.POINTER(C.c_ulonglong)]),
    ("sdm_intra_check_maps", C.c_int, [_ctx, _f32p, _f32p, _f32p]),
    ("sdm_intra_grow_maps", C.c_int, [_ctx, _f32p, _f32p, _f32p]),
    ("sdm_epipolar_search", C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.c_float, _f32p]),
    ("sdm_search_range", C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   _f32p, _f32p]),
    ("sdm_fuse", C.c_int, [_ctx, _f32p, _f32p, C.c_int, _f32p]),
    ("sdm_pair_geometry", C.c_int, [_ctx, C.c_int, C.c_int, _f32p, _f32p, _f32p]),
    ("sdm_stereo_search_constraints", C.c_int, [_f32p, C.c_int, _f32p, _f32p]),
    ("sdm_median_rot_in_plane", C.c_float, [_ip, _f32p, C.c_int, _ip, _f32p, C.c_int]),
    ("sdm_enable_stats", C.c_int, [_ctx, C.c_int]),
    ("sdm_get_stats", C.c_int, [_ctx, C.POINTER(Stats), C.c_int]),
    ("sdm_set_scan_mode", C.c_int, [_ctx, C.c_int]),
    ("sdm_get_params", C.c_int, [_ctx, C.POINTER(Params)]),
    ("sdm_set_ingest_overlap", C.c_int, [_ctx, C.c_int]),
    ("sdm_selftest", C.c_int, [_ctx, C.c_int, C.POINTER(C.c_ulonglong)]),
    ("sdm_enable_timing", C.c_int, [_ctx, C.c_int]),
    ("sdm_get_timing", C.c_int, [_ctx, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    ("sdm_device_arch", C.c_char_p, [_ctx]),
]


def load_library():
    """Loads lib/libsdm_hip.so and binds every symbol of include/sdm_c.h (AttributeError if one
    is missing).  Raises OSError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise OSError("%s not built: run `python orb-slam-free-space-carving_amd/build.py` "
                      "(there is no CPU fallback)" % p)
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def stereo_search_constraints(depths):
    lib = load_library()
    d, dp = _f32(depths)
    mn, mx = C.c_float(), C.c_float()
    rc = lib.sdm_stereo_search_constraints(dp, len(d), C.byref(mn), C.byref(mx))
    if rc:
        raise SdmError(rc, lib.sdm_last_error().decode())
    return float(mn.value), float(mx.value)


def median_rot_in_plane(mp1, ang1, mp2, ang2):
    lib = load_library()
    m1, m1p = _i32(mp1)
    m2, m2p = _i32(mp2)
    a1, a1p = _f32(ang1)
    a2, a2p = _f32(ang2)
    return float(lib.sdm_median_rot_in_plane(m1p, a1p, len(m1), m2p, a2p, len(m2)))


class Engine:
    """One context = one GPU.  Mirrors the ProbabilityMapping method surface (PM.h:72-91) at
    keyframe-slot granularity."""

    def __init__(self, W, H, max_keyframes, max_neighbours=7, device=0, batch_capacity=0,
                 with_pointset=True, ext_depth_pool=None, stream=None):
        self.lib = load_library()
        cfg = Config()
        self.lib.sdm_default_config(C.byref(cfg))
        cfg.device, cfg.W, cfg.H = device, W, H
        cfg.max_keyframes, cfg.max_neighbours = max_keyframes, max_neighbours
        cfg.batch_capacity = batch_capacity
        cfg.with_pointset = 1 if with_pointset else 0
        cfg.ext_depth_pool = ext_depth_pool
        cfg.stream = stream
        self.W, self.H, self.max_keyframes, self.max_neighbours = W, H, max_keyframes, max_neighbours
        self.ctx = _ctx()
        self._check(self.lib.sdm_create(C.byref(self.ctx), C.byref(cfg)))

    def _check(self, rc):
        if rc:
            raise SdmError(rc, self.lib.sdm_last_error().decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sdm_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------
    def arch(self):
        return self.lib.sdm_device_arch(self.ctx).decode()

    def set_params(self, **kw):
        p = Params()
        self.lib.sdm_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self._check(self.lib.sdm_set_params(self.ctx, C.byref(p)))

    def get_params(self):
        p = Params()
        self._check(self.lib.sdm_get_params(self.ctx, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Params._fields_}

    def set_stream(self, stream_ptr):
        self._check(self.lib.sdm_set_stream(self.ctx, stream_ptr))

    def synchronize(self):
        self._check(self.lib.sdm_synchronize(self.ctx))

    def depth_pool_ptr(self):
        return self.lib.sdm_depth_pool_ptr(self.ctx)

    # -- inputs ----------------------------------------------------------------------------------
    def upload_keyframe(self, slot, im, grad, theta, I_stddev, K, Tcw):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        assert im.shape == (self.H, self.W)
        g, gp = _f32(grad)
        t, tp = _f32(theta)
        k, kp = _f32(K)
        T, Tp = _f32(np.asarray(Tcw).reshape(12))
        self._check(self.lib.sdm_upload_keyframe(self.ctx, slot, im.ctypes.data_as(_u8p), gp, tp,
                                                 float(I_stddev), kp, Tp))

    def upload_image(self, slot, im, K, Tcw):
        im = np.ascontiguousarray(im, dtype=np.uint8)
        assert im.shape == (self.H, self.W)
        k, kp = _f32(K)
        T, Tp = _f32(np.asarray(Tcw).reshape(12))
        self._check(self.lib.sdm_upload_image(self.ctx, slot, im.ctypes.data_as(_u8p), kp, Tp))

    ORDER = dict(rgb=0, bgr=1, rgba=2, bgra=3, gray=4)
    compact_entries = 0  # wire format of the exchange (exchange_compact)

    def upload_image_rgb(self, slot, pixels, order, K, dist, Tcw):
        """camera frame [H, W, C] (or [H, W] gray); dist = (k1, k2, p1, p2, k3) or None"""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        ch = {0: 3, 1: 3, 2: 4, 3: 4, 4: 1}[self.ORDER[order]]
        assert px.size == self.H * self.W * ch, (px.shape, ch)
        k, kp = _f32(K)
        T, Tp = _f32(np.asarray(Tcw).reshape(12))
        dp = None
        if dist is not None:
            d, dp = _f32(dist)
            assert len(d) == 5
        self._check(self.lib.sdm_upload_image_rgb(self.ctx, slot, px.ctypes.data_as(_u8p), self.ORDER[order], kp, dp, Tp))

    def _upload_args(self, slots, images, K, Tcw, nbytes):
        n = len(slots)
        assert len(images) == n
        ims = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]  # (a pinned array stays where it is)
        for im in ims:
            assert im.size == nbytes, (im.shape, nbytes)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in ims])
        sl = (C.c_int * n)(*[int(x) for x in slots])
        k = np.ascontiguousarray(np.broadcast_to(np.asarray(K, np.float32).reshape(-1, 4), (n, 4)), dtype=np.float32)
        T = np.ascontiguousarray(np.asarray(Tcw, np.float32).reshape(n, 12))
        return n, sl, ptrs, ims, k, T

    def upload_images_batch(self, slots, images, K, Tcw):
        """n gray images [H, W] in one call; K: [4] (shared) or [n][4]; Tcw: [n] poses"""
        n, sl, ptrs, ims, k, T = self._upload_args(slots, images, K, Tcw, self.H * self.W)
        self._check(self.lib.sdm_upload_images_batch(self.ctx, n, sl, ptrs, k.ctypes.data_as(_f32p), T.ctypes.data_as(_f32p)))

    def upload_images_rgb_batch(self, slots, frames, order, K, dist, Tcw):
        ch = {0: 3, 1: 3, 2: 4, 3: 4, 4: 1}[self.ORDER[order]]
        n, sl, ptrs, ims, k, T = self._upload_args(slots, frames, K, Tcw, self.H * self.W * ch)
        dp = None
        if dist is not None:
            d, dp = _f32(dist)
            assert len(d) == 5
        self._check(self.lib.sdm_upload_images_rgb_batch(self.ctx, n, sl, ptrs, self.ORDER[order], k.ctypes.data_as(_f32p), dp,
                                                         T.ctypes.data_as(_f32p)))

    def host_alloc(self, shape, dtype=np.uint8):
        """a numpy array in pinned host memory (sdm_host_alloc); release with host_free(array)"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.sdm_host_alloc(nbytes)
        if not p:
            raise MemoryError("sdm_host_alloc(%d)" % nbytes)
        a = np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if p:
            self.lib.sdm_host_free(p)

    def upload_image_device(self, slot, dev_ptr, K, Tcw):
        k, kp = _f32(K)
        T, Tp = _f32(np.asarray(Tcw).reshape(12))
        self._check(self.lib.sdm_upload_image_device(self.ctx, slot, dev_ptr, kp, Tp))

    def set_pose(self, slot, Tcw):
        T, Tp = _f32(np.asarray(Tcw).reshape(12))
        self._check(self.lib.sdm_set_pose(self.ctx, slot, Tp))

    def download_inputs(self, slot):
        im = np.empty((self.H, self.W), np.uint8)
        g = np.empty((self.H, self.W), np.float32)
        t = np.empty((self.H, self.W), np.float32)
        s = C.c_float()
        self._check(self.lib.sdm_download_inputs(self.ctx, slot, im.ctypes.data_as(_u8p),
                                                 g.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p),
                                                 C.byref(s)))
        return im, g, t, float(s.value)

    # -- stages ------------------------------------------------------------------------------------
    def _batch_args(self, refs, nbrs, rot, min_depth, max_depth):
        refs = np.ascontiguousarray(refs, dtype=np.int32).reshape(-1)
        n_ref = len(refs)
        nbrs = np.ascontiguousarray(nbrs, dtype=np.int32).reshape(n_ref, -1)
        n = nbrs.shape[1]
        rot_a = None if rot is None else np.ascontiguousarray(rot, dtype=np.float32).reshape(n_ref, n)
        mn = np.ascontiguousarray(np.broadcast_to(np.asarray(min_depth, np.float32), (n_ref,)))
        mx = np.ascontiguousarray(np.broadcast_to(np.asarray(max_depth, np.float32), (n_ref,)))
        return refs, n_ref, nbrs, n, rot_a, mn, mx

    def _stage(self, fn, refs, nbrs, rot, min_depth, max_depth):
        refs, n_ref, nbrs, n, rot_a, mn, mx = self._batch_args(refs, nbrs, rot, min_depth, max_depth)
        self._check(fn(self.ctx, n_ref, refs.ctypes.data_as(_ip), n, nbrs.ctypes.data_as(_ip),
                       None if rot_a is None else rot_a.ctypes.data_as(_f32p),
                       mn.ctypes.data_as(_f32p), mx.ctypes.data_as(_f32p)))

    def search_fuse(self, refs, nbrs, min_depth, max_depth, rot=None):
        self._stage(self.lib.sdm_search_fuse, refs, nbrs, rot, min_depth, max_depth)

    def recon(self, refs, nbrs, min_depth, max_depth, rot=None):
        """SemiDenseRecon (PM.h:75) for a batch of reference keyframes."""
        self._stage(self.lib.sdm_recon, refs, nbrs, rot, min_depth, max_depth)

    def intra_check(self, refs):
        r, rp = _i32(np.asarray(refs).reshape(-1))
        self._check(self.lib.sdm_intra_check(self.ctx, len(r), rp))

    def intra_grow(self, refs):
        r, rp = _i32(np.asarray(refs).reshape(-1))
        self._check(self.lib.sdm_intra_grow(self.ctx, len(r), rp))

    def inter_check(self, refs, nbrs, commit=False):
        refs = np.ascontiguousarray(refs, dtype=np.int32).reshape(-1)
        nbrs = np.ascontiguousarray(nbrs, dtype=np.int32).reshape(len(refs), -1)
        self._check(self.lib.sdm_inter_check(self.ctx, len(refs), refs.ctypes.data_as(_ip), nbrs.shape[1],
                                             nbrs.ctypes.data_as(_ip), 1 if commit else 0))

    def inter_check_pointset(self, refs, nbrs, commit=False):
        """inter_check + pointset(source=1) in one call (PM.cc:300-306); one kernel for maps from SemiDenseRecon"""
        refs = np.ascontiguousarray(refs, dtype=np.int32).reshape(-1)
        nbrs = np.ascontiguousarray(nbrs, dtype=np.int32).reshape(len(refs), -1)
        self._check(self.lib.sdm_inter_check_pointset(self.ctx, len(refs), refs.ctypes.data_as(_ip), nbrs.shape[1],
                                                      nbrs.ctypes.data_as(_ip), 1 if commit else 0))

    def pointset(self, refs, source=1):
        r, rp = _i32(np.asarray(refs).reshape(-1))
        self._check(self.lib.sdm_pointset(self.ctx, len(r), rp, source))

    # -- maps ---------------------------------------------------------------------------------------
    def upload_depth(self, slot, rho, sigma):
        r, rp = _f32(rho)
        s, sp = _f32(sigma)
        assert r.shape == (self.H, self.W) and s.shape == (self.H, self.W)
        self._check(self.lib.sdm_upload_depth(self.ctx, slot, rp, sp))

    def download_depth(self, slot):
        r = np.empty((self.H, self.W), np.float32)
        s = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.sdm_download_depth(self.ctx, slot, r.ctypes.data_as(_f32p), s.ctypes.data_as(_f32p)))
        return r, s

    def download_checked(self, slot):
        r = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.sdm_download_checked(self.ctx, slot, r.ctypes.data_as(_f32p)))
        return r

    def download_pointset(self, slot):
        x = np.empty((self.H, 3 * self.W), np.float32)
        self._check(self.lib.sdm_download_pointset(self.ctx, slot, x.ctypes.data_as(_f32p)))
        return x

    def mark_depth_present(self, slots):
        r, rp = _i32(np.asarray(slots).reshape(-1))
        self._check(self.lib.sdm_mark_depth_present(self.ctx, len(r), rp))

    # -- multi-GPU exchange (RCCL inside the engine) -----------------------------------------------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        buf = (C.c_ubyte * self.COMM_ID_BYTES)()
        self._check(self.lib.sdm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid, world, rank):
        buf = (C.c_ubyte * self.COMM_ID_BYTES).from_buffer_copy(uid) if uid is not None else None
        self._check(self.lib.sdm_comm_init(self.ctx, buf, world, rank))

    def comm_destroy(self):
        self.compact_entries = 0  # sdm_comm_destroy resets the wire format to whole maps, whatever else it reports
        self._check(self.lib.sdm_comm_destroy(self.ctx))

    def comm_info(self):
        w, r = C.c_int(), C.c_int()
        self._check(self.lib.sdm_comm_info(self.ctx, C.byref(w), C.byref(r)))
        return int(w.value), int(r.value)

    def _xchg_args(self, send, recv):
        sp, ss = _i32([p for p, _ in send]), _i32([s for _, s in send])
        rp, rs = _i32([p for p, _ in recv]), _i32([s for _, s in recv])
        return (len(send), sp[1], ss[1], len(recv), rp[1], rs[1]), (sp, ss, rp, rs)

    def exchange_halo_begin(self, send, recv):
        """send / recv: lists of (peer_rank, local_slot)"""
        args, keep = self._xchg_args(send, recv)
        self._check(self.lib.sdm_exchange_halo_begin(self.ctx, *args))

    def exchange_wait(self):
        self._check(self.lib.sdm_exchange_wait(self.ctx))

    def exchange_halo(self, send, recv):
        args, keep = self._xchg_args(send, recv)
        self._check(self.lib.sdm_exchange_halo(self.ctx, *args))

    def allgather_depth(self, first_slot, count, fetch=None):
        """fetch: None = in place (slot == global keyframe index), else list of (gathered_index, local_slot)"""
        if fetch is None:
            self._check(self.lib.sdm_allgather_depth(self.ctx, first_slot, count, -1, None, None))
            return
        fi, fip = _i32([i for i, _ in fetch])
        ds, dsp = _i32([s for _, s in fetch])
        self._check(self.lib.sdm_allgather_depth(self.ctx, first_slot, count, len(fetch), fip, dsp))

    def allgather_begin(self, maps_per_rank):
        self._check(self.lib.sdm_allgather_begin(self.ctx, maps_per_rank))

    def allgather_piece(self, slots):
        s, sp = _i32(np.asarray(slots).reshape(-1))
        self._check(self.lib.sdm_allgather_piece(self.ctx, len(s), sp))

    def allgather_finish(self, fetch):
        fi, fip = _i32([i for i, _ in fetch])
        ds, dsp = _i32([s for _, s in fetch])
        self._check(self.lib.sdm_allgather_finish(self.ctx, len(fetch), fip, dsp))

    def comm_all_max(self, value):
        out = C.c_int()
        self._check(self.lib.sdm_comm_all_max(self.ctx, int(value), C.byref(out)))
        return int(out.value)

    def exchange_compact(self, entries_per_map):
        """0 = whole maps cross ranks; > 0 = the {rho,sigma} of the first entries_per_map active-list entries"""
        self._check(self.lib.sdm_exchange_compact(self.ctx, int(entries_per_map)))
        self.compact_entries = int(entries_per_map)

    COMPACT_HEADER = 8  # float2 units behind the entries (csrc/sdm_comm.h XCHG_HEADER)

    def compact_pack_host(self, slot):
        """the compact wire payload of a slot's map: float32 [entries + 8, 2]"""
        out = np.empty((self.compact_entries + self.COMPACT_HEADER, 2), np.float32)
        self._check(self.lib.sdm_compact_pack_host(self.ctx, int(slot), out.ctypes.data_as(_f32p)))
        return out

    def compact_unpack_host(self, slot, payload):
        """scatter a payload into the slot's map; returns True when it was refused (packed with another list)"""
        p = np.ascontiguousarray(payload, np.float32)
        assert p.size == 2 * (self.compact_entries + self.COMPACT_HEADER)
        ref = C.c_int()
        self._check(self.lib.sdm_compact_unpack_host(self.ctx, int(slot), p.ctypes.data_as(_f32p), C.byref(ref)))
        return bool(ref.value)

    def exchange_mismatches(self):
        out = C.c_int()
        self._check(self.lib.sdm_exchange_mismatches(self.ctx, C.byref(out)))
        return int(out.value)

    def compact_sources_ready(self, slots):
        arr = (C.c_int * max(len(slots), 1))(*[int(x) for x in slots])
        out = C.c_int()
        self._check(self.lib.sdm_compact_sources_ready(self.ctx, len(slots), arr, C.byref(out)))
        return bool(out.value)

    def active_count(self, slot):
        out = C.c_int()
        self._check(self.lib.sdm_active_count(self.ctx, int(slot), C.byref(out)))
        return int(out.value)

    def active_list(self, slot):
        """(list of y << 16 | x in raster order, 64-bit hash of the pixel set) of a slot"""
        n = self.active_count(slot)
        lst = np.empty(max(n, 1), np.uint32)
        cnt, h = C.c_int(), C.c_ulonglong()
        self._check(self.lib.sdm_download_active_list(self.ctx, int(slot), lst.ctypes.data_as(C.POINTER(C.c_uint)), int(lst.size),
                                                      C.byref(cnt), C.byref(h)))
        return lst[:cnt.value].copy(), int(h.value)

    def comm_all_ok(self, local_ok=True):
        out = C.c_int()
        self._check(self.lib.sdm_comm_all_ok(self.ctx, 1 if local_ok else 0, C.byref(out)))
        return bool(out.value)

    def assume_pipeline_maps(self, slots):
        r, rp = _i32(np.asarray(slots).reshape(-1))
        self._check(self.lib.sdm_assume_pipeline_maps(self.ctx, len(r), rp))

    def intra_check_maps(self, rho, sigma, grad=None):
        r = np.array(rho, dtype=np.float32, order="C")
        s = np.array(sigma, dtype=np.float32, order="C")
        gp = None
        if grad is not None:
            g, gp = _f32(grad)
        self._check(self.lib.sdm_intra_check_maps(self.ctx, r.ctypes.data_as(_f32p), s.ctypes.data_as(_f32p), gp))
        return r, s

    def intra_grow_maps(self, rho, sigma, grad):
        r = np.array(rho, dtype=np.float32, order="C")
        s = np.array(sigma, dtype=np.float32, order="C")
        g, gp = _f32(grad)
        self._check(self.lib.sdm_intra_grow_maps(self.ctx, r.ctypes.data_as(_f32p), s.ctypes.data_as(_f32p), gp))
        return r, s

    # -- per-pixel ------------------------------------------------------------------------------------
    def epipolar_search(self, ref, nbr, x, y, min_depth, max_depth, rot=0.0):
        out = (C.c_float * 5)()
        self._check(self.lib.sdm_epipolar_search(self.ctx, ref, nbr, x, y, min_depth, max_depth, rot, out))
        return dict(rho=float(out[0]), sigma=float(out[1]), supported=int(out[2]), best_u=float(out[3]),
                    best_v=float(out[4]))

    def search_range(self, ref, nbr, x, y, mind, maxd):
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.sdm_search_range(self.ctx, ref, nbr, x, y, mind, maxd, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def fuse(self, rho, sigma):
        r, rp = _f32(rho)
        s, sp = _f32(sigma)
        out = (C.c_float * 3)()
        self._check(self.lib.sdm_fuse(self.ctx, rp, sp, len(r), out))
        return float(out[0]), float(out[1]), int(out[2])

    def pair_geometry(self, ref, nbr):
        F = np.empty(9, np.float32)
        R = np.empty(9, np.float32)
        t = np.empty(3, np.float32)
        self._check(self.lib.sdm_pair_geometry(self.ctx, ref, nbr, F.ctypes.data_as(_f32p),
                                               R.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p)))
        return F, R, t

    # -- instrumentation ---------------------------------------------------------------------------------
    def set_ingest_overlap(self, on=True):
        """batch uploads overlap the compute calls that do not use their slots (sdm_c.h sdm_set_ingest_overlap)"""
        self._check(self.lib.sdm_set_ingest_overlap(self.ctx, 1 if on else 0))

    def set_scan_mode(self, mode):
        """0 = per wave (default), 1 = batched scan, 2 = gradient-mask scan (diagnostic; results are identical)"""
        self._check(self.lib.sdm_set_scan_mode(self.ctx, int(mode)))

    def enable_stats(self, on=True):
        self._check(self.lib.sdm_enable_stats(self.ctx, 1 if on else 0))

    def selftest(self, which):
        out = (C.c_ulonglong * 2)()
        self._check(self.lib.sdm_selftest(self.ctx, which, out))
        return int(out[0]), int(out[1])

    STAGES = ("search_fuse", "intra", "inter", "pointset")

    def enable_timing(self, on=True):
        self._check(self.lib.sdm_enable_timing(self.ctx, 1 if on else 0))

    def get_timing(self, reset=True):
        """{stage: (total_ms, launches)} from HIP events on the engine's stream"""
        ms = (C.c_double * 4)()
        cnt = (C.c_longlong * 4)()
        self._check(self.lib.sdm_get_timing(self.ctx, ms, cnt, 1 if reset else 0))
        return {s: (float(ms[i]), int(cnt[i])) for i, s in enumerate(self.STAGES)}

    def get_stats(self, reset=True):
        s = Stats()
        self._check(self.lib.sdm_get_stats(self.ctx, C.byref(s), 1 if reset else 0))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}
