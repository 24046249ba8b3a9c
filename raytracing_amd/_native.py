"""ctypes binding of libt3d_hip.so (the C ABI declared in include/t3d.h).

There is no fallback: if the library is missing or no MI355X is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .flatten import FlatScene, Opts, SceneDesc, Stats, make_opts

LIB_PATH = os.environ.get("T3D_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libt3d_hip.so")
ABI_VERSION = 9
BRUTE_FORCE_MAX_PRIMS = 176   # T3D_BRUTE_FORCE_MAX_PRIMS of include/t3d.h: scenes above it go through the LBVH (accel 'auto')

SYMBOLS = ("t3d_abi_version", "t3d_device_count", "t3d_create", "t3d_destroy", "t3d_last_error",
           "t3d_upload_scene", "t3d_update_scene", "t3d_render_rows", "t3d_render_rows_device", "t3d_trace_pixel", "t3d_trace_ray",
           "t3d_closest_hits", "t3d_last_call_stats", "t3d_set_timing", "t3d_set_accel", "t3d_set_host_tree_limit", "t3d_set_background_tree", "t3d_tree_status", "t3d_set_rounds", "t3d_set_cancel",
           "t3d_render_rows_async", "t3d_host_alloc", "t3d_host_free", "t3d_render_frame_multi",
           "t3d_render_rows_device_static", "t3d_graph_counters", "t3d_render_rows_device_split", "t3d_render_rows_async_split",
           "t3d_post_upscale_device", "t3d_post_denoise_device", "t3d_post_upscale", "t3d_post_denoise")


class NativeError(RuntimeError):
    """The HIP library is missing, failed to load, or a call into it failed (``code``: the T3D_E_* value)."""
    code = None


class Cancelled(NativeError):
    """The cancel callback asked the render call to stop (T3D_E_CANCELLED)."""


class Node(C.Structure):
    """ctypes mirror of ``t3d_node``."""
    _fields_ = [("heap", C.c_int32), ("obj", C.c_int32), ("face", C.c_int32), ("direction", C.c_int32),
                ("t", C.c_float), ("point", C.c_float * 3), ("normal", C.c_float * 3),
                ("ray_start", C.c_float * 3), ("ray_dir", C.c_float * 3), ("ray_ior", C.c_float),
                ("ray_obj", C.c_int32), ("color", C.c_float * 3)]


CANCEL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

E_CAPACITY = -4
E_CANCELLED = -5

_lib = None


def _share_hip_runtime_with_torch():
    """A PyTorch-ROCm wheel carries its own libamdhip64.so.7.  A process can hold one HIP runtime only: whichever copy
    is loaded first serves both users, and torch finds no GPU when it is made to run on the system's copy.  So when
    such a wheel is installed (it need not be imported, and is not imported here) its copy is loaded first; without
    one, the library's own run path (/opt/rocm/lib) applies."""
    if os.environ.get("T3D_SYSTEM_HIP_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def load():
    """Load libt3d_hip.so once and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} not found: build it with `make -C raytracing_amd/csrc` "
                          "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    P = C.POINTER
    lib.t3d_abi_version.restype = C.c_int
    lib.t3d_device_count.restype = C.c_int
    lib.t3d_create.restype = C.c_int
    lib.t3d_create.argtypes = [C.c_int, P(C.c_void_p)]
    lib.t3d_destroy.restype = None
    lib.t3d_destroy.argtypes = [C.c_void_p]
    lib.t3d_last_error.restype = C.c_char_p
    lib.t3d_last_error.argtypes = [C.c_void_p]
    lib.t3d_upload_scene.restype = C.c_int
    lib.t3d_upload_scene.argtypes = [C.c_void_p, P(SceneDesc)]
    lib.t3d_update_scene.restype = C.c_int
    lib.t3d_update_scene.argtypes = [C.c_void_p, P(SceneDesc)]
    lib.t3d_render_rows.restype = C.c_int
    lib.t3d_render_rows.argtypes = [C.c_void_p, P(Opts), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, P(Stats)]
    lib.t3d_render_rows_device.restype = C.c_int
    lib.t3d_render_rows_device.argtypes = [C.c_void_p, P(Opts), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, P(Stats)]
    lib.t3d_render_rows_device_static.restype = C.c_int
    lib.t3d_render_rows_device_static.argtypes = lib.t3d_render_rows_device.argtypes
    lib.t3d_trace_pixel.restype = C.c_int
    lib.t3d_trace_pixel.argtypes = [C.c_void_p, P(Opts), C.c_int, C.c_int, C.c_int, P(Node), C.c_int]
    lib.t3d_trace_ray.restype = C.c_int
    lib.t3d_trace_ray.argtypes = [C.c_void_p, P(Opts), P(C.c_float), P(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                  P(Node), C.c_int]
    lib.t3d_closest_hits.restype = C.c_int
    lib.t3d_closest_hits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.t3d_last_call_stats.restype = C.c_int
    lib.t3d_last_call_stats.argtypes = [C.c_void_p, P(Stats)]
    lib.t3d_set_timing.restype = C.c_int
    lib.t3d_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_set_rounds.restype = C.c_int
    lib.t3d_set_rounds.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_set_accel.restype = C.c_int
    lib.t3d_set_accel.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_set_host_tree_limit.restype = C.c_int
    lib.t3d_set_host_tree_limit.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_set_background_tree.restype = C.c_int
    lib.t3d_set_background_tree.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_tree_status.restype = C.c_int
    lib.t3d_tree_status.argtypes = [C.c_void_p, C.c_int]
    lib.t3d_render_rows_async.restype = C.c_int
    lib.t3d_render_rows_async.argtypes = [C.c_void_p, P(Opts), C.c_int, C.c_int, C.c_void_p]
    lib.t3d_render_frame_multi.restype = C.c_int
    lib.t3d_render_frame_multi.argtypes = [P(C.c_void_p), C.c_int, P(Opts), C.c_void_p, C.c_int, P(Stats)]
    lib.t3d_host_alloc.restype = C.c_int
    lib.t3d_host_alloc.argtypes = [C.c_size_t, P(C.c_void_p)]
    lib.t3d_host_free.restype = C.c_int
    lib.t3d_host_free.argtypes = [C.c_void_p]
    lib.t3d_render_rows_device_split.restype = C.c_int
    lib.t3d_render_rows_device_split.argtypes = [P(C.c_void_p), C.c_int, P(Opts), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.t3d_render_rows_async_split.restype = C.c_int
    lib.t3d_render_rows_async_split.argtypes = [P(C.c_void_p), C.c_int, P(Opts), C.c_int, C.c_int, C.c_void_p]
    lib.t3d_graph_counters.restype = C.c_int
    lib.t3d_graph_counters.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    lib.t3d_set_cancel.restype = C.c_int
    lib.t3d_set_cancel.argtypes = [C.c_void_p, CANCEL_FN, C.c_void_p]
    lib.t3d_post_upscale_device.restype = C.c_int
    lib.t3d_post_upscale_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.t3d_post_denoise_device.restype = C.c_int
    lib.t3d_post_denoise_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.t3d_post_upscale.restype = C.c_int
    lib.t3d_post_upscale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.t3d_post_denoise.restype = C.c_int
    lib.t3d_post_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    if lib.t3d_abi_version() != ABI_VERSION:
        raise NativeError(f"libt3d_hip.so has ABI version {lib.t3d_abi_version()}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def device_count() -> int:
    n = load().t3d_device_count()
    if n < 0:
        raise NativeError("t3d_device_count failed: no usable HIP runtime")
    return n


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class _PinnedBlock:
    """One page-locked host allocation; goes back to the pool when the last array that views it is collected."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        if load().t3d_host_alloc(nbytes, C.byref(p)) != 0:
            raise NativeError(f"t3d_host_alloc({nbytes}) failed")
        self.ptr, self.nbytes = p.value, nbytes
        self.buffer = (C.c_uint8 * nbytes).from_address(self.ptr)

    def free(self):
        if self.ptr:
            load().t3d_host_free(C.c_void_p(self.ptr))
            self.ptr = None


_pinned_pool = {}     # nbytes -> [free blocks]
_PINNED_KEEP = 4      # free blocks kept per size; more are released


def pinned_image(shape):
    """A uint8 array of ``shape`` in page-locked host memory (the device copies into it at the link's rate, and
    asynchronously).  The memory returns to a small pool when the array and every view of it are gone."""
    import weakref
    nbytes = int(np.prod(shape))
    free = _pinned_pool.setdefault(nbytes, [])
    block = free.pop() if free else _PinnedBlock(nbytes)
    root = np.frombuffer(block.buffer, dtype=np.uint8, count=nbytes)

    def give_back(b=block, n=nbytes):
        lst = _pinned_pool.setdefault(n, [])
        if len(lst) < _PINNED_KEEP:
            lst.append(b)
        else:
            b.free()
    # every view (the reshaped array returned here, a channel flip of it, ...) keeps `root` alive, so `root` dies
    # last: that is when the block is free again
    weakref.finalize(root, give_back)
    return root.reshape(shape)


class Context:
    """One device context (``t3d_ctx``) holding one uploaded scene."""

    def __init__(self, device: int = 0):
        self._lib = load()
        handle = C.c_void_p()
        rc = self._lib.t3d_create(int(device), C.byref(handle))
        if rc != 0:
            raise NativeError(f"t3d_create(device={device}) failed ({rc}): "
                              f"{self._lib.t3d_last_error(None).decode()}")
        self._h = handle
        self.device = device
        self.flat = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.t3d_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            err = (Cancelled if rc == E_CANCELLED else NativeError)(f"{what} failed ({rc}): {self._lib.t3d_last_error(self._h).decode()}")
            err.code = rc
            raise err
        return rc

    def upload(self, flat: FlatScene):
        desc = flat.as_desc()
        self._check(self._lib.t3d_upload_scene(self._h, C.byref(desc)), "t3d_upload_scene")
        self.flat = flat

    def update(self, flat: FlatScene):
        """The uploaded scene with spheres, Lights or the camera moved (``t3d_update_scene``): no grouping, no build."""
        desc = flat.as_desc()
        self._check(self._lib.t3d_update_scene(self._h, C.byref(desc)), "t3d_update_scene")
        self.flat = flat

    def render_rows(self, options: dict, seed: int = 0, row0: int = 0, nrows: int | None = None,
                    want_radiance=False, want_primary=False, row_stride: int = 1, out=None, frames: int = 1, frame_seed_step: int = 0):
        """-> dict(bgr uint8 [rows,W,3], radiance, prim [rows,W,2], prim_t, stats dict); ``out``: a C-contiguous
        uint8 array [rows,W,3] (e.g. a block of rows of the caller's image) that receives the pixels in place.
        ``frames`` > 1: the rows are rendered that many times in one call, frame f with the seed ``seed + f *
        frame_seed_step``; every output then has a leading axis of that length"""
        opts = make_opts(options, seed, row_stride=row_stride, frames=frames, frame_seed_step=frame_seed_step)
        W = opts.width
        nrows = (opts.height - row0 + row_stride - 1) // row_stride if nrows is None else nrows
        lead = (frames, nrows) if frames > 1 else (nrows,)
        if out is None:
            bgr = np.empty(lead + (W, 3), np.uint8)
        else:
            bgr = out
            if bgr.dtype != np.uint8 or bgr.shape != lead + (W, 3) or not bgr.flags.c_contiguous or not bgr.flags.writeable:
                raise ValueError(f"out must be a writable C-contiguous uint8 array of shape {lead + (W, 3)}")
        rad = np.empty(lead + (W, 3), np.float32) if want_radiance else None
        prim = np.empty(lead + (W, 2), np.int32) if want_primary else None
        prim_t = np.empty(lead + (W,), np.float32) if want_primary else None
        stats = Stats()
        self._check(self._lib.t3d_render_rows(self._h, C.byref(opts), row0, nrows, _ptr(bgr), _ptr(rad), _ptr(prim),
                                              _ptr(prim_t), C.byref(stats)), "t3d_render_rows")
        return dict(bgr=bgr, radiance=rad, prim=prim, prim_t=prim_t, stats=stats.as_dict())

    def render_rows_async(self, options: dict, seed: int, row0: int, nrows: int, out, row_stride: int = 1):
        """Start rendering into ``out`` (a ``pinned_image`` of shape [nrows, W, 3]) and return at once;
        ``last_call_stats()`` waits for the frame and returns its counters."""
        opts = make_opts(options, seed, row_stride=row_stride)
        if out.dtype != np.uint8 or out.shape != (nrows, opts.width, 3) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {(nrows, opts.width, 3)}")
        self._check(self._lib.t3d_render_rows_async(self._h, C.byref(opts), row0, nrows, _ptr(out)), "t3d_render_rows_async")

    def render_rows_device(self, options: dict, seed: int, row0: int, nrows: int, bgr_ptr: int, radiance_ptr: int = 0,
                           primary_ptr: int = 0, primary_t_ptr: int = 0, stream: int = 0, want_stats=True,
                           row_stride: int = 1, static=False, frames: int = 1, frame_seed_step: int = 0):
        """Outputs go to caller-owned device memory (raw addresses, e.g. ``tensor.data_ptr()``).  ``static``: through
        ``t3d_render_rows_device_static``, which allocates nothing (work buffers in the library's per-GPU static array).
        ``frames`` > 1: the outputs hold that many frames of ``nrows`` rows, one after the other (``make_opts``)."""
        opts = make_opts(options, seed, row_stride=row_stride, frames=frames, frame_seed_step=frame_seed_step)
        stats = Stats() if want_stats else None
        fn = self._lib.t3d_render_rows_device_static if static else self._lib.t3d_render_rows_device
        self._check(fn(
            self._h, C.byref(opts), row0, nrows, C.c_void_p(bgr_ptr), C.c_void_p(radiance_ptr or None),
            C.c_void_p(primary_ptr or None), C.c_void_p(primary_t_ptr or None), C.c_void_p(stream or None),
            C.byref(stats) if stats is not None else None), "t3d_render_rows_device_static" if static else "t3d_render_rows_device")
        return stats.as_dict() if stats is not None else None

    def last_call_stats(self):
        """Synchronise the most recent (possibly asynchronous) render call and return its stats."""
        stats = Stats()
        self._check(self._lib.t3d_last_call_stats(self._h, C.byref(stats)), "t3d_last_call_stats")
        return stats.as_dict()

    def set_accel(self, mode):
        """'auto' | 'brute' | 'lbvh' for scenes uploaded after this call."""
        code = {"auto": 0, "brute": 1, "lbvh": 2}[mode]
        self._check(self._lib.t3d_set_accel(self._h, code), "t3d_set_accel")

    def set_host_tree_limit(self, max_prims: int):
        """Scenes of up to this many primitives, uploaded after this call, get the hierarchy that is cut on the host by
        surface area (better, slower to build); larger ones the device's Morton-order one.  0: always the device."""
        self._check(self._lib.t3d_set_host_tree_limit(self._h, int(max_prims)), "t3d_set_host_tree_limit")

    def set_background_tree(self, on: bool):
        """Scenes uploaded after this call whose hierarchy the device builds are traced through it at once while a host
        thread cuts the surface-area tree, which the first call that finds it finished puts in its place."""
        self._check(self._lib.t3d_set_background_tree(self._h, 1 if on else 0), "t3d_set_background_tree")

    TREE_STATUS = {0: "none", 1: "device", 2: "host", 3: "building", 4: "background"}

    def tree_status(self, wait: bool = False) -> str:
        """Where the uploaded scene's hierarchy comes from: 'none' | 'device' | 'host' | 'building' | 'background'.
        ``wait``: first wait for a background build and adopt its tree."""
        rc = self._lib.t3d_tree_status(self._h, 1 if wait else 0)
        self._check(min(rc, 0), "t3d_tree_status")
        return self.TREE_STATUS[rc]

    def set_full_rounds(self, full: bool):
        """Always 128 rays per trace round (for callers that keep several bands in flight) instead of the adaptive cut."""
        self._check(self._lib.t3d_set_rounds(self._h, 1 if full else 0), "t3d_set_rounds")

    def set_cancel(self, poll):
        """``poll()`` -> truthy to stop: asked before every band and after every tree level of a render call, which then
        raises ``Cancelled``.  ``None`` removes the callback.  An exception inside ``poll`` counts as "stop"."""
        if poll is None:
            self._cancel_fn = CANCEL_FN()
        else:
            def trampoline(_user):
                try:
                    return 1 if poll() else 0
                except Exception:
                    return 1
            self._cancel_fn = CANCEL_FN(trampoline)    # kept alive for as long as the library may call it
        self._check(self._lib.t3d_set_cancel(self._h, self._cancel_fn, None), "t3d_set_cancel")

    def post_upscale(self, image, factor: int):
        """``Post.upscale`` on the GPU: uint8 [H, W, C] -> uint8 [H * factor, W * factor, C]."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim != 3:
            raise ValueError("image must have shape [height, width, channels]")
        h, w, ch = img.shape
        out = np.empty((h * max(int(factor), 0), w * max(int(factor), 0), ch), np.uint8)
        self._check(self._lib.t3d_post_upscale(self._h, _ptr(img), h, w, ch, int(factor), _ptr(out)), "t3d_post_upscale")
        return out

    def post_denoise(self, image, kernel_size: int):
        """``Post.denoise`` on the GPU: uint8 [H, W, 3] -> uint8 [H, W, 3] (per-channel median, zero border)."""
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("image must have shape [height, width, 3]")
        h, w, _ = img.shape
        out = np.empty_like(img)
        self._check(self._lib.t3d_post_denoise(self._h, _ptr(img), h, w, int(kernel_size), _ptr(out)), "t3d_post_denoise")
        return out

    def post_upscale_device(self, in_ptr: int, height: int, width: int, channels: int, factor: int, out_ptr: int, stream: int = 0):
        """Device-pointer form (raw addresses); queued on ``stream`` (0: the context's) without waiting."""
        self._check(self._lib.t3d_post_upscale_device(self._h, C.c_void_p(in_ptr or None), height, width, channels, factor,
                                                      C.c_void_p(out_ptr or None), C.c_void_p(stream or None)), "t3d_post_upscale_device")

    def post_denoise_device(self, in_ptr: int, height: int, width: int, kernel_size: int, out_ptr: int, stream: int = 0):
        """Device-pointer form (raw addresses); queued on ``stream`` (0: the context's) without waiting."""
        self._check(self._lib.t3d_post_denoise_device(self._h, C.c_void_p(in_ptr or None), height, width, kernel_size,
                                                      C.c_void_p(out_ptr or None), C.c_void_p(stream or None)), "t3d_post_denoise_device")

    def graph_counters(self):
        """Launch graphs of repeated bands on this context: how many were captured, how often one was replayed, and how
        often a replay first had its seed changed (``t3d_graph_counters``)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._lib.t3d_graph_counters(self._h, C.byref(a), C.byref(b), C.byref(c)), "t3d_graph_counters")
        return {"captured": a.value, "replayed": b.value, "reseeded": c.value}

    def set_timing(self, per_kernel: bool):
        """Per-kernel event timing on/off (on: ms_k_level / ms_resolve filled, a few us of idle per launch)."""
        self._check(self._lib.t3d_set_timing(self._h, int(bool(per_kernel))), "t3d_set_timing")

    def trace_pixel(self, options: dict, x: int, y: int, sample: int = 0, seed: int = 0):
        opts = make_opts(options, seed)
        cap = min(65536, (1 << (opts.bounces + 1)) - 1)
        buf = (Node * cap)()
        n = self._check(self._lib.t3d_trace_pixel(self._h, C.byref(opts), x, y, sample, buf, cap), "t3d_trace_pixel")
        return [buf[i] for i in range(n)]

    def trace_ray(self, options: dict, origin, direction, ior=1.0, inside_object=-1, key=(0, 0, 0), seed: int = 0):
        """Tree of one explicit ray; ``options['bounces']`` / ``['lights']`` are the arguments of ``Photon.trace``."""
        opts = make_opts(options, seed)
        cap = min(65536, (1 << (opts.bounces + 1)) - 1)
        buf = (Node * cap)()
        o = (C.c_float * 3)(*[float(c) for c in origin])
        d = (C.c_float * 3)(*[float(c) for c in direction])
        n = self._check(self._lib.t3d_trace_ray(self._h, C.byref(opts), o, d, float(ior), int(inside_object), int(key[0]),
                                                int(key[1]), int(key[2]), buf, cap), "t3d_trace_ray")
        return [buf[i] for i in range(n)]

    def closest_hits(self, origins, dirs, lights=True, back_cull=True):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        obj, face, dirflag = (np.empty(n, np.int32) for _ in range(3))
        t = np.empty(n, np.float32)
        normal = np.empty((n, 3), np.float32)
        self._check(self._lib.t3d_closest_hits(self._h, n, _ptr(o), _ptr(d), int(lights), int(back_cull), _ptr(obj),
                                               _ptr(face), _ptr(t), _ptr(dirflag), _ptr(normal)), "t3d_closest_hits")
        return dict(obj=obj, face=face, t=t, direction=dirflag, normal=normal)


def render_rows_split(contexts, options: dict, seed: int, row0: int, nrows: int, bgr_ptr: int = 0, stream: int = 0, out=None,
                      row_stride: int = 1):
    """The rows of one call as ``len(contexts)`` contiguous bands, band k traced by ``contexts[k]`` (contexts of one
    device holding the same scene), all at once: shorter latency for a single frame.  Not waited for.  Output: the device
    address ``bgr_ptr`` (``t3d_render_rows_device_split``, on ``stream``) or ``out``, a ``pinned_image`` of shape
    [nrows, W, 3] (``t3d_render_rows_async_split``).  ``split_call_stats(contexts)`` waits and returns the counters."""
    lib = load()
    n = len(contexts)
    opts = make_opts(options, seed, row_stride=row_stride)
    handles = (C.c_void_p * n)(*[c._h for c in contexts])
    if out is not None:
        if out.dtype != np.uint8 or out.shape != (nrows, opts.width, 3) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {(nrows, opts.width, 3)}")
        rc = lib.t3d_render_rows_async_split(handles, n, C.byref(opts), row0, nrows, _ptr(out))
        contexts[0]._check(rc, "t3d_render_rows_async_split")
    else:
        rc = lib.t3d_render_rows_device_split(handles, n, C.byref(opts), row0, nrows, C.c_void_p(bgr_ptr or None), C.c_void_p(stream or None))
        contexts[0]._check(rc, "t3d_render_rows_device_split")


def split_call_stats(contexts):
    """Waits for a ``render_rows_split`` call and returns its counters: the sum over the bands (times: the slowest band's).
    Raises ``NativeError`` (code E_CAPACITY) if a band's trace trees did not fit its buffers -- after every context has
    been asked, so that none is left with an unreported overflow."""
    total, error = None, None
    for c in contexts:
        try:
            st = c.last_call_stats()
        except NativeError as e:
            error = error or e
            continue
        if total is None:
            total = st
        else:
            for k, v in st.items():
                if isinstance(v, list):
                    total[k] = [max(a, b) if k == "level_ms" else a + b for a, b in zip(total[k], v)]
                elif k.startswith("ms_"):
                    total[k] = max(total[k], v)
                else:
                    total[k] += v
    if error is not None:
        raise error
    return total


def render_frame_multi(contexts, options: dict, seed: int, out=None, frame_ptr: int = 0):
    """One frame on several contexts (normally one per GPU), rows interleaved: context k renders rows k, k + n, ...
    Every context must hold the same scene.  ``out``: uint8 [H, W, 3] host array (``pinned_image`` for full speed);
    or ``frame_ptr``: address of a device buffer of that size.  -> stats dict (counters summed over the contexts)."""
    lib = load()
    n = len(contexts)
    opts = make_opts(options, seed)
    if out is not None:
        if out.dtype != np.uint8 or out.shape != (opts.height, opts.width, 3) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {(opts.height, opts.width, 3)}")
    elif not frame_ptr:
        raise ValueError("give a host array or a device address")
    handles = (C.c_void_p * n)(*[c._h for c in contexts])
    stats = Stats()
    rc = lib.t3d_render_frame_multi(handles, n, C.byref(opts), _ptr(out) if out is not None else C.c_void_p(frame_ptr),
                                    0 if out is not None else 1, C.byref(stats))
    contexts[0]._check(rc, "t3d_render_frame_multi")
    return stats.as_dict()
