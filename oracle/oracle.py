"""ctypes front-end of the CPU oracle (oracle/libt3d_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline
leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd.flatten import FlatScene, Opts, SceneDesc, Stats, make_opts

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libt3d_oracle.so")


class ONode(C.Structure):
    _fields_ = [("px", C.c_int32), ("py", C.c_int32), ("sample", C.c_int32), ("heap", C.c_int32),
                ("obj", C.c_int32), ("face", C.c_int32), ("direction", C.c_int32), ("ray_obj", C.c_int32),
                ("t", C.c_double), ("ray_ior", C.c_double),
                ("point", C.c_double * 3), ("normal", C.c_double * 3), ("ray_start", C.c_double * 3),
                ("ray_dir", C.c_double * 3), ("color", C.c_double * 3)]


NODE_DTYPE = np.dtype([("px", "<i4"), ("py", "<i4"), ("sample", "<i4"), ("heap", "<i4"), ("obj", "<i4"),
                       ("face", "<i4"), ("direction", "<i4"), ("ray_obj", "<i4"), ("t", "<f8"),
                       ("ray_ior", "<f8"), ("point", "<f8", 3), ("normal", "<f8", 3), ("ray_start", "<f8", 3),
                       ("ray_dir", "<f8", 3), ("color", "<f8", 3)])
assert NODE_DTYPE.itemsize == C.sizeof(ONode)


def build(force=False):
    """Compile the oracle with its Makefile (gcc, a second or two)."""
    # make decides: the library depends on the C file and on both headers (struct layouts)
    subprocess.check_call(["make", "-s", "-C", HERE] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.t3do_keyed_bits24.restype = C.c_uint32
        _lib.t3do_keyed_bits24.argtypes = [C.c_uint32] * 6
        _lib.t3do_attenuation.restype = C.c_double
        _lib.t3do_attenuation.argtypes = [C.c_double]
        _lib.t3do_py_floordiv.restype = C.c_double
        _lib.t3do_py_floordiv.argtypes = [C.c_double, C.c_double]
    return _lib


def _p(arr, ctype):
    return None if arr is None else arr.ctypes.data_as(C.POINTER(ctype))


def render(flat: FlatScene, options: dict, seed=0, rng_mode=0, row0=0, nrows=None, want_nodes=False,
           want_radiance=True, want_fragility=False, node_cap=None, row_stride=1):
    """Render rows with the oracle.  Returns a dict: bgr, rgb, radiance, prim (obj, face), prim_t,
    stats, nodes (structured array) ..."""
    opts = make_opts(options, seed, rng_mode, row_stride)
    W, H = opts.width, opts.height
    nrows = (H - row0 + row_stride - 1) // row_stride if nrows is None else nrows
    desc = flat.as_desc()
    bgr = np.zeros((nrows, W, 3), np.uint8)
    rad = np.zeros((nrows, W, 3), np.float64) if want_radiance else None
    prim = np.zeros((nrows, W, 2), np.int32)
    prim_t = np.zeros((nrows, W), np.float64)
    frag = np.zeros((nrows, W), np.float64) if want_fragility else None
    stats = Stats()
    nodes = None
    written = C.c_int64(0)
    if want_nodes:
        cap = node_cap or max(1024, nrows * W * max(1, opts.samples) * 8)
        nodes = np.zeros(cap, NODE_DTYPE)
    rc = lib().t3do_render_rows(C.byref(desc), C.byref(opts), row0, nrows, _p(bgr, C.c_uint8), _p(rad, C.c_double),
                                _p(prim, C.c_int32), _p(prim_t, C.c_double), _p(frag, C.c_double),
                                C.byref(stats), _p(nodes, ONode) if nodes is not None else None,
                                0 if nodes is None else len(nodes), C.byref(written))
    if rc != 0:
        raise RuntimeError(f"t3do_render_rows failed: {rc}")
    out = dict(bgr=bgr, rgb=bgr[:, :, ::-1].copy(), radiance=rad, prim=prim, prim_t=prim_t, fragility=frag,
               stats=stats.as_dict())
    if nodes is not None:
        if written.value > len(nodes):
            raise RuntimeError("oracle node buffer too small")
        out["nodes"] = nodes[:written.value]
    return out


def primary_fragility(flat: FlatScene, options: dict, row0=0, nrows=None, row_stride=1):
    opts = make_opts(options, row_stride=row_stride)
    nrows = (opts.height - row0 + row_stride - 1) // row_stride if nrows is None else nrows
    desc = flat.as_desc()
    frag = np.zeros((nrows, opts.width), np.float64)
    rc = lib().t3do_primary_fragility(C.byref(desc), C.byref(opts), row0, nrows, _p(frag, C.c_double))
    if rc != 0:
        raise RuntimeError(f"t3do_primary_fragility failed: {rc}")
    return frag


def first_collision(flat: FlatScene, start, direction, lights=True, back_cull=True, options=None):
    opts = make_opts(options or dict(width=1, height=1))
    desc = flat.as_desc()
    s = (C.c_double * 3)(*start)
    d = (C.c_double * 3)(*direction)
    obj, face, dirflag = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
    t = C.c_double(0)
    point, normal = (C.c_double * 3)(), (C.c_double * 3)()
    lib().t3do_first_collision(C.byref(desc), C.byref(opts), s, d, int(lights), int(back_cull), C.byref(obj),
                               C.byref(face), C.byref(t), point, normal, C.byref(dirflag))
    return dict(obj=obj.value, face=face.value, t=t.value, point=list(point), normal=list(normal),
                direction=dirflag.value)


def face_collision(start, direction, verts):
    t, dirflag = C.c_double(0), C.c_int32(0)
    point, normal = (C.c_double * 3)(), (C.c_double * 3)()
    flatv = [c for v in verts for c in v]
    hit = lib().t3do_face_collision((C.c_double * 3)(*start), (C.c_double * 3)(*direction), (C.c_double * 9)(*flatv),
                                    C.byref(t), point, normal, C.byref(dirflag))
    return dict(t=t.value, point=list(point), normal=list(normal), direction=dirflag.value) if hit else None


def sphere_collision(start, direction, center, radius, is_light=False, back_cull=True):
    t, dirflag = C.c_double(0), C.c_int32(0)
    point, normal = (C.c_double * 3)(), (C.c_double * 3)()
    hit = lib().t3do_sphere_collision((C.c_double * 3)(*start), (C.c_double * 3)(*direction),
                                      (C.c_double * 3)(*center), C.c_double(radius), int(is_light), int(back_cull),
                                      C.byref(t), point, normal, C.byref(dirflag))
    return dict(t=t.value, point=list(point), normal=list(normal), direction=dirflag.value) if hit else None


def enters_aabb(start, direction, bmin, bmax):
    return bool(lib().t3do_enters_aabb((C.c_double * 3)(*start), (C.c_double * 3)(*direction),
                                       (C.c_double * 3)(*bmin), (C.c_double * 3)(*bmax)))


def bvh_leaf_order(flat: FlatScene, obj: int):
    desc = flat.as_desc()
    n = int(flat.obj_tri_count[obj])
    out = np.zeros(max(n, 1), np.int32)
    sizes = np.zeros(max(n, 1), np.int32)
    nl = C.c_int32(0)
    k = lib().t3do_bvh_leaf_order(C.byref(desc), obj, _p(out, C.c_int32), len(out), _p(sizes, C.c_int32), len(sizes),
                                  C.byref(nl))
    return out[:k].tolist(), sizes[:nl.value].tolist()


def object_box(flat: FlatScene, obj: int):
    desc = flat.as_desc()
    lo, hi = (C.c_double * 3)(), (C.c_double * 3)()
    lib().t3do_object_box(C.byref(desc), obj, lo, hi)
    return list(lo), list(hi)


def camera_ray(flat: FlatScene, options: dict, px: int, py: int):
    opts = make_opts(options)
    desc = flat.as_desc()
    pv, s, d = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_double * 3)()
    lib().t3do_camera_ray(C.byref(desc), C.byref(opts), px, py, pv, s, d)
    return list(pv), list(s), list(d)


def mt_uniforms(seed, n, a=-1.0, b=1.0):
    out = np.zeros(n, np.float64)
    lib().t3do_mt_uniforms(C.c_uint32(seed), n, C.c_double(a), C.c_double(b), _p(out, C.c_double))
    return out
