"""``Photon`` -- the ray-tracing render engine -- and ``Rendered``, its result object.

Same call surface as the reference's ``T3DR`` module (T3DR.py:23-118 ``NoBoundCameraError`` /
``Rendered``, T3DR.py:501-1242 ``Photon``, T3DR.py:1247-1267 ``ThreadsProgress``), but all tracing
happens in the HIP library ``libt3d_hip.so`` on an MI355X; there is no CPU rendering path.

Differences a caller can observe, all deliberate:

* ``threads`` no longer creates Python threads.  It still has to divide ``height`` (same
  ``ValueError``) and it is the number of stripes between which ``progressCallback`` fires and
  ``cancelCallback`` is polled -- per stripe, not per pixel.
* random directions for bounce()/transmit() come from the keyed generator in ``keyed_rng`` (option
  ``seed``, default 0), so an image is reproducible and independent of how rows are split.
* ``Rendered.traces`` is a lazy mapping: ``traces[(x, y)]`` re-traces that pixel on the GPU.
* with ``lights`` False an emitter hit by the initial ray is still a leaf of the trace tree (the reference traces
  children below it that ``computeColor`` never looks at, T3DR.py:973, 1014-1041): same pixels and ``rays``, smaller
  trees from ``trace`` / ``traces``.
* extra option keys: ``device`` (HIP device index, default 0), ``seed``, ``gpus`` (render the frame on that many
  devices at once, rows interleaved; must divide ``height``), ``devices`` (which ones) and ``bands`` (a frame rendered
  without callbacks is traced as that many bands of rows at once on the one device, which shortens its latency; default:
  two for small scenes and large frames, else one).
"""
from __future__ import annotations

import copy
import json
import threading
import time as _time
from collections.abc import Mapping

import numpy as np

from . import _native
from .flatten import FlattenCache, flatten_scene
from .scene import RGB, Ray, Scale, TraceTreeNode, Vec3


class NoBoundCameraError(Exception):
    def __init__(self, message="Scene contains no bound camera. Bind a camera in order to render.\n\n"
                               "\t\tScene().camera = Camera()\n\n\tOR\tcamera = Camera(Scene())\n\n"):
        self.message = message
        super().__init__(self.message)


class ThreadsProgress:
    """Lock-guarded progress table; every update calls ``callback(table, key, total)``
    (T3DR.py:1247-1267)."""

    def __init__(self, callback, total):
        self.lock = threading.Lock()
        self.total = total
        self.callback = callback
        self._table = {}

    def __getitem__(self, key):
        with self.lock:
            return self._table[key]

    def __setitem__(self, key, value):
        with self.lock:
            self._table[key] = value
            self.callback(self._table, key, self.total)


class LazyTraces(Mapping):
    """``{(x, y): [TraceTreeNode | None per sample]}`` computed on demand by re-tracing the pixel."""

    def __init__(self, photon, width, height, samples, options, seed):
        self._photon, self._w, self._h, self._s = photon, width, height, samples
        self._options, self._seed = dict(options), seed
        self._cache = {}

    def __len__(self):
        return self._w * self._h

    def __iter__(self):
        return ((x, y) for y in range(self._h) for x in range(self._w))

    def __contains__(self, key):
        return (isinstance(key, tuple) and len(key) == 2 and 0 <= key[0] < self._w and 0 <= key[1] < self._h)

    def __getitem__(self, key):
        if key not in self:
            raise KeyError(key)
        if key not in self._cache:
            self._cache[key] = [self._photon._trace_pixel_tree(self._options, key[0], key[1], s, self._seed)
                                for s in range(self._s)]
        return self._cache[key]


class Rendered:
    """A finished image plus how it was made (T3DR.py:30-118).  ``imgData`` is RGB uint8 [H, W, 3]."""

    def __init__(self, imgData, traces, width, height, time=None, rays=None, options=None, failed=False, engine=None):
        self.imgData = np.flip(np.asarray(imgData), 2)  # callers hand over BGR
        self.traces = traces
        self.width = width
        self.height = height
        self.time = time
        self.rays = rays
        self.options = options
        self.failed = failed
        self.engine = engine
        self.stats = None

    def imageAs(self, type):
        if type == "np":
            return self.imgData
        if type == "cv2":
            return np.flip(self.imgData, 2)
        if type in ("listList", "list"):
            return self.imgData.astype(int).tolist()
        if type == "tupleList":
            return [[tuple(int(c) for c in pix) for pix in row] for row in self.imgData]
        return None

    def toDict(self, includeTraces=False):
        options = dict(self.options or {})
        if "ambient" in options:
            options["ambient"] = [*options["ambient"]]
        traces = {}
        if includeTraces:
            traces = {f"{x},{y}": [t.toDict() if t is not None else None for t in per_pixel]
                      for (x, y), per_pixel in self.traces.items()}
        return {"imgData": self.imageAs("listList"), "traces": traces, "width": self.width, "height": self.height,
                "time": self.time, "rays": self.rays, "options": options, "failed": self.failed, "engine": self.engine}

    @staticmethod
    def fromDict(data):
        options = data.get("options", {})
        if "ambient" in options:
            options["ambient"] = RGB(*options["ambient"])
        rgb = np.asarray(data.get("imgData", []), dtype=np.uint8).reshape(-1, max(1, data.get("width", 1440)), 3)
        traces = {tuple(int(v) for v in key.split(",")): [TraceTreeNode.fromDict(t) if t is not None else None for t in per_pixel]
                  for key, per_pixel in data.get("traces", {}).items()}
        return Rendered(np.flip(rgb, 2), traces, data.get("width", 1440), data.get("height", 900), data.get("time", 0),
                        data.get("rays", 0), options, failed=data.get("failed", False), engine=data.get("engine", "debug"))

    def saveJSON(self, filepath):
        with open(filepath, "w") as f:
            json.dump(self.toDict(), f)

    @staticmethod
    def fromJSON(filepath):
        with open(filepath, "r") as f:
            return Rendered.fromDict(json.load(f))


_DEFAULTS = {"debug": False, "bounces": 1, "wait": True, "samples": 1, "bvh": True, "debugOptions": None,
             "aabb": True, "lights": True, "transformBefore": True, "threads": 4, "width": 1440, "height": 900}


class Photon:
    """Ray tracer.  ``Photon(scene, options).render()`` -> ``Rendered``."""

    def __init__(self, scene, options=None):
        self._ctx = None
        self._flat = None
        self._baked = None
        self._uploaded_digest = None
        self.uploads = 0           # scene uploads so far (a re-flattening with an unchanged digest is not one)
        self.updates = 0           # of which in place: only spheres, Lights or the camera had moved
        self._multi = []           # [context, digest of its scene] per GPU when options['gpus'] > 1
        self._multi_devices = None
        self._helpers = []         # [context, digest of its scene]: trace the other bands of a frame (options['bands'])
        self.scene = scene
        self.options = self.mergeDefaultOptions({} if options is None else options)
        self.rays = 0

    # -- scene / options -----------------------------------------------------------------------
    @property
    def scene(self):
        return self._scene

    @scene.setter
    def scene(self, scene):
        self._scene = scene
        self._flat = None
        self._baked = None
        self._cache = FlattenCache()

    @property
    def transformedScene(self):
        """Deep copy of the scene with mesh transforms baked in (T3DR.py:1090-1100), made on first
        use; before any render it is the scene itself, as in the reference (T3DR.py:504)."""
        if self._baked is None:
            return self._scene
        if self._baked is True:
            baked = copy.deepcopy(self._scene)
            for obj in baked.objects:
                if obj.type == "mesh":
                    obj.setTransforms()
            self._baked = baked
        return self._baked

    @transformedScene.setter
    def transformedScene(self, scene):
        self._baked = scene

    def mergeDefaultOptions(self, options):
        """Fill the missing keys of ``options`` in place (T3DR.py:519-525)."""
        for key, value in _DEFAULTS.items():
            if key not in options:
                options[key] = {} if key == "debugOptions" else value
        if "ambient" not in options:
            options["ambient"] = RGB(20, 20, 20)
        return options

    def _camera_scene(self):
        return self.transformedScene if self.options["transformBefore"] else self._scene

    def _new_context(self, device=None):
        """A context on ``device`` (default: the option of that name) set up as the options ask: ``hostTreeLimit``, absent from
        the reference, is up to how many primitives a scene's hierarchy is cut on the host by surface area rather than on
        the device in Morton order (``t3d_set_host_tree_limit``; worth raising for a large scene rendered many times)."""
        ctx = _native.Context(int(self.options.get("device", 0) if device is None else device))
        if self.options.get("hostTreeLimit") is not None:
            ctx.set_host_tree_limit(int(self.options["hostTreeLimit"]))
        if self.options.get("backgroundTree"):
            # also absent from the reference: a scene above that limit is traced through the device's tree at once while a
            # host thread cuts the better one, adopted by the first call that finds it finished (t3d_set_background_tree;
            # for a scene that is rendered again and again, e.g. with a moving camera)
            ctx.set_background_tree(True)
        return ctx

    def _context(self):
        if self._ctx is None:
            self._ctx = self._new_context()
        return self._ctx

    def _prepare(self, force_flatten=False):
        """Flatten the scene and upload it if it changed.  Live-edited scenes (``transformBefore`` False, the
        picking path of explorer.py:324-355) are re-flattened on every call, and so is the scene of every
        ``render()`` (which may have been edited since the last one)."""
        if self._scene.camera is None:
            raise NoBoundCameraError()
        if force_flatten or self._flat is None or not self.options["transformBefore"]:
            # only what changed since the last look is baked again; a render compares the meshes entry by entry, a query
            # that looks at a live scene before every ray (picking) goes by the objects' signatures
            before = self._flat.digest() if self._flat is not None else None
            flat = flatten_scene(self._scene, self._cache, thorough=force_flatten or self._flat is None)
            digest = flat.digest()
            # an unchanged scene keeps its device copy (plane groups, hierarchy); one in which only spheres, Lights or the
            # camera moved has it updated in place (no grouping, no hierarchy build)
            if digest != self._uploaded_digest:
                if self._cache.change == "moved" and before is not None and before == self._uploaded_digest:
                    self._context().update(flat)
                    self.updates += 1
                else:
                    self._context().upload(flat)
                self._uploaded_digest = digest
                self.uploads += 1
            self._flat = flat
        elif self._uploaded_digest is None:
            # flattened by the several-GPU path, whose contexts are its own: this one has seen no scene yet
            self._context().upload(self._flat)
            self._uploaded_digest = self._flat.digest()
            self.uploads += 1
        return self._context()

    def _bands(self):
        """How many bands of rows a plain render() traces at once: the launches of one 1080p frame of a small scene leave
        part of the device empty at their starts and ends, which a second band's launches fill."""
        o = self.options
        if o.get("bands"):
            return max(1, int(o["bands"]))
        flat = self._flat
        small_scene = flat is not None and flat.n_tris + flat.n_spheres <= _native.BRUTE_FORCE_MAX_PRIMS
        samples = int(o["samples"]) if int(o["bounces"]) > 1 else 1
        return 2 if small_scene and int(o["width"]) * int(o["height"]) * samples >= 500_000 and int(o["height"]) >= 16 else 1

    def _band_contexts(self):
        """The contexts of a banded render, every one holding the scene of ``self._ctx``; just that one when the helpers
        cannot be used."""
        want = self._bands() - 1
        if want <= 0 or self._flat is None or self._uploaded_digest is None:
            return [self._ctx]
        while len(self._helpers) < want:
            self._helpers.append([self._new_context(), None])
        for slot in self._helpers[:want]:
            if slot[1] != self._uploaded_digest:
                slot[0].upload(self._flat)
                slot[1] = self._uploaded_digest
        return [self._ctx] + [slot[0] for slot in self._helpers[:want]]

    def _native_options(self):
        o = self.options
        return {"width": o["width"], "height": o["height"], "bounces": o["bounces"], "samples": o["samples"],
                "lights": o["lights"], "bvh": o["bvh"], "aabb": o["aabb"], "ambient": tuple(o["ambient"])}

    # -- camera (T3DR.py:528-551) ----------------------------------------------------------------
    def pixelToVector(self, pixelX, pixelY) -> Vec3:
        camera = self._camera_scene().camera
        width, height = self.options["width"], self.options["height"]
        ratio = width / height
        local = Vec3((pixelX / width) * ratio - ratio / 2, 0.5 - pixelY / height, camera.length / 20)
        return local.applyTransforms(camera.position, Scale(1, 1, 1), camera.rotation)

    def getRay(self, pixelVec) -> Ray:
        start = self._camera_scene().camera.position
        return Ray(start, (pixelVec - start).normalise(), 1, None)

    # -- single queries ----------------------------------------------------------------------------
    def getFirstCollision(self, ray, lights, backCull=True):
        """Closest hit of one ray (T3DR.py:654-730); dict with None entries for a miss."""
        ctx = self._prepare()
        hit = ctx.closest_hits([[*ray.start]], [[*ray.direction]], lights, backCull)
        oi = int(hit["obj"][0])
        if oi < 0:
            return {"object": None, "point": None, "t": None, "normal": None, "direction": None, "face": None}
        obj = self._camera_scene().objects[oi]
        t = float(hit["t"][0])
        fi = int(hit["face"][0])
        return {"object": obj, "point": ray.pointOnRay(t), "t": t, "normal": Vec3(*(float(c) for c in hit["normal"][0])),
                "direction": int(hit["direction"][0]), "face": obj.mesh.faces[fi] if fi >= 0 else None}

    def getAttenuation(self, distance, k=0.01) -> float:
        return max(0, min(1, 1 / (1 + k * distance ** 2)))

    def getAllLightsPos(self):
        """``[[position, strength, colour], ...]`` in the order shading visits them (T3DR.py:789-816)."""
        flat = self._flat if (self._flat is not None and self.options["transformBefore"]) else flatten_scene(self._scene)
        return [[Vec3(*map(float, p)), float(s), RGB(*map(float, c))]
                for p, s, c in zip(flat.light_pos, flat.light_strength, flat.light_color)]

    def getVisibility(self, ray) -> float:
        """1 if the first thing the ray meets is a light / emissive face (or nothing), else
        1 - reflectivity of the blocker (T3DR.py:762-778)."""
        hit = self.getFirstCollision(ray, True, False)
        obj = hit["object"]
        if obj is None or obj.type == "light":
            return 1
        shader = obj.getShader(hit["face"])
        if obj.type == "mesh" and shader.emissionStrength > 0:
            return 1
        return 1 - shader.reflectivity

    def getDirectLight(self, collisionInfo):
        """``(light colour, intensity)`` at a hit (T3DR.py:820-900): per light a diffuse and a mirror term, the
        shadow ray's visibility and the distance attenuation.  The shadow rays of all lights are traced by the
        device in one call; ``rays`` grows by one per light as in the reference."""
        point, obj = collisionInfo["point"], collisionInfo["object"]
        lights = self.getAllLightsPos()
        total_color, total = RGB(0, 0, 0), 0
        if not lights:
            return total_color, total
        shader = obj.getShader(collisionInfo["face"])
        rough = shader.roughness if shader else 1
        normal = collisionInfo["normal"].normalise()
        view = (self._camera_scene().camera.position - point).normalise()
        rays = []
        for pos, _, _ in lights:
            to_light = (pos - point).normalise()
            rays.append(Ray(point + to_light * 0.2, to_light, 1, None))
        self.rays += len(rays)
        vis = self._visibilities(rays)
        for (pos, strength, color), ray, v in zip(lights, rays, vis):
            lv = ray.direction.normalise()
            diffuse = max(0, lv.dot(normal))
            mirror = (normal * lv.dot(normal) * 2) - lv
            specular = 0 if mirror.normalise().dot(view) < 0.96 else 1
            att = self.getAttenuation((pos - point).magnitude())
            total += (diffuse * rough + specular * (1 - rough)) * strength * v * att
            total_color = total_color + color * 0.6
        return total_color, total

    def _visibilities(self, rays):
        """getVisibility (T3DR.py:762-778) for a batch of rays, one device call."""
        ctx = self._prepare()
        hit = ctx.closest_hits([[*r.start] for r in rays], [[*r.direction] for r in rays], True, False)
        objects = self._camera_scene().objects
        out = []
        for oi, fi in zip(hit["obj"], hit["face"]):
            if oi < 0 or objects[oi].type == "light":
                out.append(1)
                continue
            obj = objects[int(oi)]
            shader = obj.getShader(obj.mesh.faces[int(fi)] if fi >= 0 else None)
            out.append(1 if (obj.type == "mesh" and shader.emissionStrength > 0) else 1 - shader.reflectivity)
        return out

    def trace(self, ray, bounces, lights=False):
        """Trace tree of one ray (T3DR.py:959-1001), built by the device (``t3d_trace_ray``).  Every node
        carries its resolved colour, so ``computeColor`` of such a tree costs nothing.  The diffuse
        directions are keyed by ``ray.key = (x, y, sample)`` when the ray has one, else (0, 0, 0)."""
        ctx = self._prepare()
        objects = self._camera_scene().objects
        inside = -1
        if ray.object is not None:
            for k, o in enumerate(objects):
                if o is ray.object or o.id == getattr(ray.object, "id", None):
                    inside = k
                    break
        opts = dict(self._native_options(), bounces=int(bounces), lights=bool(lights))
        nodes = ctx.trace_ray(opts, ray.start, ray.direction, ray.ior, inside, getattr(ray, "key", (0, 0, 0)),
                              int(self.options.get("seed", 0)))
        self.rays += 1
        return self._tree_from_nodes(nodes, objects)

    def computeColor(self, trace):
        """Colour of a trace tree, bottom-up with a clamp on every addition (T3DR.py:1004-1083).  Trees that
        came from the device already hold the colour of every node; a tree assembled by the caller is
        evaluated here, its shadow rays on the device."""
        if trace is None:
            return self.options["ambient"]
        info = trace.collisionInfo
        if "color" in info:
            return RGB(*info["color"])
        obj = info["object"]
        camera = self._camera_scene().camera
        ambient = self.options["ambient"]
        if obj.type == "light":
            att = self.getAttenuation((info["point"] - camera.position).magnitude())
            return (obj.shaders[0].color * obj.strength * att).clamp()
        shader = obj.getShader(info["face"])
        if obj.type == "mesh" and shader.emissionStrength > 0:
            att = self.getAttenuation((info["point"] - camera.position).magnitude())
            return (shader.color * shader.emissionStrength * att).clamp()
        slot = []
        for child in (trace.left, trace.right):
            if child is None:
                slot.append((ambient * (2 / 255), self.getAttenuation(0)))
            else:
                slot.append((self.computeColor(child),
                             self.getAttenuation((child.collisionInfo["point"] - trace.ray.start).magnitude())))
        refl = shader.reflectivity
        indirect = (slot[0][0] * refl * slot[0][1]) + (slot[1][0] * (1 - refl) * slot[1][1] * 1.4)
        light_color, intensity = self.getDirectLight(info)
        direct = ((shader.color * intensity) + (light_color * intensity * (2 / 255))) * refl
        return (indirect + direct + (shader.color * ambient * (1 / 255))).clamp()

    def _trace_pixel_tree(self, options, x, y, sample, seed):
        ctx = self._prepare()
        nodes = ctx.trace_pixel(options, x, y, sample, seed)
        return self._tree_from_nodes(nodes, self.transformedScene.objects)

    def _tree_from_nodes(self, nodes, objects):
        if not nodes:
            return None
        by_heap = {}
        for n in nodes:
            obj = objects[n.obj]
            inside = objects[n.ray_obj] if n.ray_obj >= 0 else None
            ray = Ray(Vec3(*n.ray_start), Vec3(*n.ray_dir), n.ray_ior, inside)
            info = {"object": obj, "point": Vec3(*n.point), "t": n.t, "normal": Vec3(*n.normal),
                    "direction": n.direction, "face": obj.mesh.faces[n.face] if n.face >= 0 else None,
                    "color": RGB(*n.color)}
            by_heap[n.heap] = TraceTreeNode(ray, info)
        for heap, node in by_heap.items():
            node.left = by_heap.get(2 * heap)
            node.right = by_heap.get(2 * heap + 1)
        return by_heap.get(1)

    # -- several GPUs -------------------------------------------------------------------------------------
    def _render_multi(self, gpus, progressCallback, cancelCallback, returnType):
        """``options['gpus'] = N``: the frame's rows are dealt to N device contexts (rows k, k + N, ... to context k),
        which render side by side and copy their rows into the frame (``t3d_render_frame_multi``).  Stands where the
        reference starts ``threads`` stripe threads and stacks their stripes (T3DR.py:1111-1165); like there, N must
        divide the image height.  ``options['devices']`` may list the HIP device of every context (default 0..N-1).
        Callbacks fire once, around the whole frame."""
        opts = self.options
        width, height = int(opts["width"]), int(opts["height"])
        if height % gpus != 0:
            raise ValueError("Number of GPUs must be a factor of image height")
        devices = [int(d) for d in (opts.get("devices") or range(gpus))]
        if len(devices) != gpus:
            raise ValueError("options['devices'] must list one device per GPU")
        if self._multi_devices != devices:
            for c, _ in self._multi:
                c.close()
            self._multi = [[self._new_context(d), None] for d in devices]
            self._multi_devices = devices
        flat = flatten_scene(self._scene, self._cache, thorough=True)
        digest = flat.digest()
        for slot in self._multi:
            if slot[1] != digest:
                slot[0].upload(flat)
                slot[1] = digest
                self.uploads += 1
        self._flat, self._baked = flat, True
        if digest != self._uploaded_digest:
            # the context that serves traces / picking still holds an older scene: _prepare uploads this one on first use
            self._uploaded_digest = None
        native_opts = self._native_options()
        seed = int(opts.get("seed", 0))
        progress = ThreadsProgress(callback=progressCallback or (lambda table, key, total: None), total=height * width)
        self.threadsProgress = progress
        progress[0] = 0
        image = _native.pinned_image((height, width, 3))
        start = _time.time()
        failed = bool(cancelCallback and cancelCallback())
        totals = {}
        if not failed:
            totals = _native.render_frame_multi([c for c, _ in self._multi], native_opts, seed, out=image)
            progress[0] = height * width - 1
            failed = bool(cancelCallback and cancelCallback())
        elapsed = _time.time() - start
        if returnType == "np":
            return image
        samples = int(opts["samples"]) if int(opts["bounces"]) > 1 else 1
        rendered = Rendered(image, LazyTraces(self, width, height, samples, native_opts, seed), width, height, elapsed,
                            int(totals.get("rays_api", 0)), opts, failed=failed, engine="photon")
        rendered.stats = totals
        return rendered

    # -- render (T3DR.py:1086-1242) ------------------------------------------------------------------
    def render(self, progressCallback=None, cancelCallback=None, returnType="rendered"):
        """Render the frame.  The reference deep-copies the scene on every call, so a scene edited between two
        calls must show; here the scene is flattened again on every call and uploaded only when its fingerprint
        changed.  When a device copy exists and nobody asked for callbacks, the device starts on the frame at once
        with that copy while the host does the flattening; the frame is kept if the fingerprint turns out
        unchanged and rendered again otherwise."""
        if self._scene.camera is None:
            raise NoBoundCameraError()
        opts = self.options
        width, height, stripes = int(opts["width"]), int(opts["height"]), int(opts["threads"])
        gpus = int(opts.get("gpus", 1) or 1)
        if height % stripes != 0:
            raise ValueError("Number of threads must be a factor of image height")
        if gpus > 1:
            return self._render_multi(gpus, progressCallback, cancelCallback, returnType)
        native_opts = self._native_options()
        seed = int(opts.get("seed", 0))
        plain = progressCallback is None and cancelCallback is None
        if plain:
            stripes = 1
        progressCallback = progressCallback or (lambda table, key, total: None)
        cancelCallback = cancelCallback or (lambda: False)
        stripe_rows = height // stripes
        progress = ThreadsProgress(callback=progressCallback, total=stripe_rows * width)
        self.threadsProgress = progress
        image = _native.pinned_image((height, width, 3))
        totals = {}
        failed = False
        self.rays = 0
        start = _time.time()
        early = None
        if plain and self._ctx is not None and self._uploaded_digest is not None:
            try:
                early = self._band_contexts()
                if len(early) > 1:
                    _native.render_rows_split(early, native_opts, seed, 0, height, out=image)
                else:
                    self._ctx.render_rows_async(native_opts, seed, 0, height, image)
            except _native.NativeError:
                # e.g. an overflow left by an earlier call: the checked path below deals with it; a helper may still hold one
                for c in (early or [])[1:]:
                    try:
                        c.last_call_stats()
                    except _native.NativeError:
                        pass
                early = None
        uploads_before = self.uploads
        ctx = self._prepare(force_flatten=True)
        self._baked = True  # transformedScene is rebuilt lazily from the current scene
        if early and self.uploads == uploads_before:      # the scene the device started on is still the scene
            try:
                totals = _native.split_call_stats(early)   # waits for the frame started above
            except _native.NativeError as e:
                if e.code != _native.E_CAPACITY:
                    raise
                early = None                          # the trees outgrew the buffers: render again, checked
        else:
            if early:
                # the frame started on the old device scene is thrown away -- and with it whatever it has to report: an
                # overflow of its node buffers must not be handed to the checked call below as if it were its own
                try:
                    _native.split_call_stats(early)
                except _native.NativeError as e:
                    if e.code != _native.E_CAPACITY:
                        raise
            early = None
        if not early:
            if not plain:
                image[:] = 0                          # a cancelled render leaves the rest of the (recycled) buffer black
                ctx.set_cancel(cancelCallback)
            try:
                for sid in range(stripes):
                    progress[sid] = 0
                    if failed:
                        continue
                    # a stripe is a block of whole rows of the image: the library writes it in place
                    try:
                        out = ctx.render_rows(native_opts, seed=seed, row0=sid * stripe_rows, nrows=stripe_rows,
                                              out=image[sid * stripe_rows:(sid + 1) * stripe_rows])
                    except _native.Cancelled:
                        failed = True
                        continue
                    for k, v in out["stats"].items():
                        if isinstance(v, list):      # per-level arrays
                            totals[k] = [a + b for a, b in zip(totals.get(k, [0] * len(v)), v)]
                        else:
                            totals[k] = totals.get(k, 0) + v
                    progress[sid] = stripe_rows * width - 1
                    if cancelCallback():
                        failed = True
            finally:
                if not plain:
                    ctx.set_cancel(None)
        elapsed = _time.time() - start
        rays = int(totals.get("rays_api", 0))
        self.rays = 0
        if returnType == "np":
            return image
        samples = int(opts["samples"]) if int(opts["bounces"]) > 1 else 1
        traces = LazyTraces(self, width, height, samples, native_opts, seed)
        rendered = Rendered(image, traces, width, height, elapsed, rays, opts, failed=failed, engine="photon")
        rendered.stats = totals
        return rendered
