"""Scene model of the T3D engine: the user-facing classes a script builds a scene with.

API-compatible with the reference's ``T3DE`` module (class, attribute and method names, JSON
schema), written from scratch.  None of the per-ray arithmetic of the renderer lives here any
more -- ``raytracing_amd.flatten`` turns a ``Scene`` into flat arrays for the HIP library -- but
the small vector classes keep their full algebra because user scripts use them.

Reference anchors: math types T3DE.py:15-270 and T3DE.py:463-604, RGB T3DE.py:365-460, scene
graph T3DE.py:607-822 and T3DE.py:952-1282, BVH T3DE.py:879-949.
"""
from __future__ import annotations

import copy as _copy
import json as _json
import math as _math
import random as _random
import struct as _struct

_NUM = (int, float)
_EPS = 1e-6


class _Triple:
    """Shared plumbing of the three-component value types (iteration, indexing, printing)."""

    __slots__ = ("x", "y", "z")
    _open, _close = "(", ")"

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    def __iter__(self):
        yield self.x
        yield self.y
        yield self.z

    def __getitem__(self, i):
        return (self.x, self.y, self.z)[i]

    def __setitem__(self, i, value):
        setattr(self, "xyz"[i], value)

    def __repr__(self):
        return f"{self._open}{self.x}, {self.y}, {self.z}{self._close}"

    def __format__(self, spec):
        return (f"{self._open}{format(self.x, spec)}, {format(self.y, spec)}, "
                f"{format(self.z, spec)}{self._close}")


class Matrix9:
    """Row-major 3x3 matrix (T3DE.py:16-45)."""

    def __init__(self, elements):
        self.elements = elements

    def __mul__(self, other):
        m = self.elements
        if type(other) is Vec3:
            return Vec3(*(m[r] * other.x + m[r + 1] * other.y + m[r + 2] * other.z for r in (0, 3, 6)))
        if type(other) is Matrix9:
            o = other.elements
            return Matrix9([m[r] * o[c] + m[r + 1] * o[c + 3] + m[r + 2] * o[c + 6]
                            for r in (0, 3, 6) for c in (0, 1, 2)])
        raise TypeError("Matrix9 can only be multiplied by Vec3 or another Matrix9")


class Euler(_Triple):
    """XYZ Euler angles in radians; the matrix is Rz*Ry*Rx (T3DE.py:49-94)."""

    __slots__ = ()
    _open, _close = "◿ ", " ◺"

    def __init__(self, x=0, y=0, z=0):
        super().__init__(x, y, z)

    def __add__(self, other):
        if type(other) is not Euler:
            raise TypeError(f"Cannot add type {type(other)} to Euler")
        return Euler(self.x + other.x, self.y + other.y, self.z + other.z)

    def __mul__(self, k):
        if type(k) not in _NUM:
            raise TypeError(f"Cannot multiply Euler with type {type(k)}")
        return Euler(self.x * k, self.y * k, self.z * k)

    def toMatrix(self):
        cx, sx = _math.cos(self.x), _math.sin(self.x)
        cy, sy = _math.cos(self.y), _math.sin(self.y)
        cz, sz = _math.cos(self.z), _math.sin(self.z)
        rx = Matrix9([1, 0, 0, 0, cx, -sx, 0, sx, cx])
        ry = Matrix9([cy, 0, sy, 0, 1, 0, -sy, 0, cy])
        rz = Matrix9([cz, -sz, 0, sz, cz, 0, 0, 0, 1])
        return rz * ry * rx


class Scale(_Triple):
    """Per-axis scale factor (T3DE.py:825-876)."""

    __slots__ = ()
    _open, _close = "❰", "❱"

    def _combine(self, other, op, verb):
        if type(other) in _NUM:
            return Scale(op(self.x, other), op(self.y, other), op(self.z, other))
        if type(other) is Scale:
            return Scale(op(self.x, other.x), op(self.y, other.y), op(self.z, other.z))
        raise TypeError(f"Cannot {verb} type {type(other)}")

    def __mul__(self, other):
        return self._combine(other, lambda a, b: a * b, "multiply Scale with")

    def __truediv__(self, other):
        return self._combine(other, lambda a, b: a / b, "divide Scale by")


class Vec3(_Triple):
    """3-D vector (T3DE.py:464-604)."""

    __slots__ = ()
    _open, _close = "⟦", "⟧"

    def __add__(self, o):
        return Vec3(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3(self.x - o.x, self.y - o.y, self.z - o.z)

    def __neg__(self):
        return Vec3(-self.x, -self.y, -self.z)

    def __mul__(self, o):
        if type(o) in _NUM:
            return Vec3(self.x * o, self.y * o, self.z * o)
        if type(o) in (Vec3, Scale):
            return Vec3(self.x * o.x, self.y * o.y, self.z * o.z)
        raise TypeError(f"Cannot multiply Vec3 with type {type(o)}")

    # scalar * vector: the reference needs it inside project() (T3DE.py:510) but never defined it
    __rmul__ = __mul__

    def __truediv__(self, o):
        if type(o) in _NUM:
            return Vec3(self.x / o, self.y / o, self.z / o)
        if type(o) is Vec3:
            return Vec3(self.x / o.x, self.y / o.y, self.z / o.z)
        raise TypeError(f"Cannot divide Vec3 by type {type(o)}")

    @staticmethod
    def _need_vec(o, what):
        if type(o) is not Vec3:
            raise TypeError(f"Cannot {what} type {type(o)}")

    def dot(self, o):
        self._need_vec(o, "dot Vec3 against")
        return self.x * o.x + self.y * o.y + self.z * o.z

    def cross(self, o):
        self._need_vec(o, "cross Vec3 against")
        return Vec3(self.y * o.z - self.z * o.y, self.z * o.x - self.x * o.z, self.x * o.y - self.y * o.x)

    def magnitude(self):
        return _math.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2)

    def normalise(self):
        return self / self.magnitude()

    def component(self, o):
        self._need_vec(o, "get component of Vec3 along")
        return self.dot(o.normalise())

    def project(self, o):
        self._need_vec(o, "project Vec3 onto")
        unit = o.normalise()
        return unit * self.component(unit)

    def flip(self, direction):
        self._need_vec(direction, "flip Vec3 across")
        return self - self.project(direction.normalise()) * 2

    def perpendicular(self, o):
        return -_EPS < self.dot(o) < _EPS

    def parallel(self, o):
        return -_EPS < self.cross(o).magnitude() < _EPS

    def angle(self, o):
        return _math.acos(self.dot(o) / (self.magnitude() * o.magnitude()))

    def applyTransforms(self, position=None, scale=None, rotation=None):
        """rotate (Rz*Ry*Rx), then scale per world axis, then translate (T3DE.py:530-544)."""
        position = Vec3(0, 0, 0) if position is None else position
        scale = Scale(0, 0, 0) if scale is None else scale
        rotation = Euler(0, 0, 0) if rotation is None else rotation
        return (rotation.toMatrix() * self) * scale + position


class Ray:
    """Half line with the medium it travels in (T3DE.py:125-247)."""

    def __init__(self, start, direction, ior, object):
        self.start = start
        self.direction = direction.normalise()
        self.ior = ior
        self.object = object

    def pointOnRay(self, t):
        return self.start + self.direction * t

    def toVec3(self):
        return (self.pointOnRay(1) - self.start).normalise()

    def getTPlane(self, plane):
        n = plane.normal
        return (n.dot(plane.position) - n.dot(self.start)) / n.dot(self.direction)

    def getTSphere(self, sphere):
        rel = self.start - sphere.position
        b = rel.dot(self.direction)
        disc = b ** 2 - (rel.dot(rel) - sphere.radius ** 2)
        if disc < 0:
            return (None, None)
        root = _math.sqrt(disc)
        return (-b - root, -b + root)

    def entersAABB(self, box):
        lo, hi = box
        t_in, t_out = float("-inf"), float("inf")
        for axis in range(3):
            o, d = self.start[axis], self.direction[axis]
            if d == 0:
                if o < lo[axis] or o > hi[axis]:
                    return False
                continue
            a, b = (lo[axis] - o) / d, (hi[axis] - o) / d
            if a > b:
                a, b = b, a
            t_in, t_out = max(t_in, a), min(t_out, b)
        return t_in <= t_out and t_out > 0

    def toDict(self):
        return {"start": [*self.start], "direction": [*self.direction], "ior": self.ior,
                "object": self.object.toDict() if self.object else None}

    @staticmethod
    def fromDict(data):
        obj = data.get("object")
        return Ray(Vec3(*data.get("start", [0, 0, 0])), Vec3(*data.get("direction", [0, 0, 0])),
                   data.get("ior", 1), Object.fromDict(obj) if obj else None)

    def __repr__(self):
        return f"<{self.start}, {self.direction}>"

    def __format__(self, spec):
        return f"<{format(self.start, spec)}, {format(self.direction, spec)}>"


class MathematicalPlane:
    """Plane through a face: anchor vertex + un-normalised normal (T3DE.py:251-263)."""

    def __init__(self, vertices):
        self.position = vertices[0]
        self.normal = (vertices[1] - vertices[0]).cross(vertices[2] - vertices[0])


class MathematicalSphere:
    def __init__(self, position, radius):
        self.position = position
        self.radius = radius


class TraceTreeNode:
    """Node of the binary trace tree: left = reflected ray, right = transmitted (T3DE.py:273-362)."""

    def __init__(self, ray, collisionInfo, left=None, right=None):
        self.ray = ray
        self.collisionInfo = collisionInfo
        self.left = left
        self.right = right

    def toDict(self):
        ci = self.collisionInfo
        face = None if ci.get("face") is None else [int(v) for v in ci["face"]]
        return {
            "ray": self.ray.toDict(),
            # the hit position is stored under 'point' by the tracer; 'coordinate' is the key the
            # reference's serialiser used
            "collisionInfo": {"coordinate": [*ci.get("coordinate", ci.get("point"))],
                              "object": ci["object"].toDict(), "normal": [*ci["normal"]],
                              "direction": int(ci["direction"]), "face": face},
            "left": self.left.toDict() if self.left else None,
            "right": self.right.toDict() if self.right else None,
        }

    @staticmethod
    def fromDict(data):
        ci = data.get("collisionInfo") or {}
        face = ci.get("face")
        point = Vec3(*ci["coordinate"]) if ci.get("coordinate") is not None else None
        info = {"coordinate": point, "point": point,
                "object": Object.fromDict(ci["object"]) if ci.get("object") else None,
                "normal": Vec3(*ci["normal"]) if ci.get("normal") is not None else None,
                "direction": ci.get("direction"), "face": None if face is None else tuple(face)}
        ray = Ray.fromDict(data.get("ray", {"start": [0, 0, 1], "direction": [0, 0, 1]}))
        left, right = data.get("left"), data.get("right")
        return TraceTreeNode(ray, info, TraceTreeNode.fromDict(left) if left else None,
                             TraceTreeNode.fromDict(right) if right else None)

    def _lines(self, depth, tag):
        ci = self.collisionInfo
        where = ci.get("point", ci.get("coordinate"))
        obj = ci["object"]
        yield (" | " * depth + tag + f"⟬{self.ray:.2f}, ({where:.2f}, {(obj.name, obj.type)}, "
               f"{ci['normal']:.2f}, {ci['direction']}, {ci['face']})⟭")
        for child, t in ((self.left, "0: "), (self.right, "1: ")):
            if child is not None:
                yield from child._lines(depth + 1, t)

    def __repr__(self):
        return "\n".join(self._lines(0, "")) + "\n"


class RGB:
    """Colour with 0..255 channels; addition saturates at 255 (T3DE.py:366-460)."""

    _default = tuple(_random.randint(0, 255) for _ in range(3))  # unspecified channels: fixed at import

    def __init__(self, r=None, g=None, b=None):
        self.r = self._default[0] if r is None else r
        self.g = self._default[1] if g is None else g
        self.b = self._default[2] if b is None else b

    def clamp(self):
        return RGB(min(self.r, 255), min(self.g, 255), min(self.b, 255))

    def __add__(self, o):
        if type(o) is not RGB:
            raise TypeError(f"Cannot add RGB to type {type(o)}")
        return RGB(self.r + o.r, self.g + o.g, self.b + o.b).clamp()

    def __mul__(self, o):
        if type(o) in _NUM:
            return RGB(self.r * o, self.g * o, self.b * o)
        if type(o) is RGB:
            return RGB(self.r * o.r, self.g * o.g, self.b * o.b)
        raise TypeError(f"Cannot multiply RGB with type {type(o)}")

    def __truediv__(self, o):
        if type(o) in _NUM:
            return RGB(self.r / o, self.g / o, self.b / o)
        if type(o) is RGB:
            return RGB(self.r / o.r, self.g / o.g, self.b / o.b)
        raise TypeError(f"Cannot divide RGB by type {type(o)}")

    @staticmethod
    def mean(colors):
        n = len(colors)
        return RGB(sum(c.r for c in colors) // n, sum(c.g for c in colors) // n,
                   sum(c.b for c in colors) // n).clamp()

    def toHex(self):
        return f"#{self.r:02x}{self.g:02x}{self.b:02x}"

    def toTuple(self, space="rgb"):
        order = {"rgb": (self.r, self.g, self.b), "bgr": (self.b, self.g, self.r),
                 "rgba": (self.r, self.g, self.b, 255), "bgra": (self.b, self.g, self.r, 255)}
        if space not in order:
            raise TypeError(f"Color space {space} not recognised")
        return order[space]

    def __iter__(self):
        return iter((self.r, self.g, self.b))

    def __repr__(self):
        return f"【{self.r}, {self.g}, {self.b}】"

    def __format__(self, spec):
        if spec == "h":
            return self.toHex()
        return f"【{format(self.r, spec)}, {format(self.g, spec)}, {format(self.b, spec)}】"


class Shader:
    """Surface properties of a face or sphere (T3DE.py:694-726)."""

    def __init__(self, name=None, debugColor=None, color=None, roughness=1.0, reflectivity=1.0,
                 emissionStrength=0.0):
        self.name = "New shader" if name is None else name
        self.debugColor = RGB(200, 200, 200) if debugColor is None else debugColor
        self.color = RGB(200, 200, 200) if color is None else color
        self.roughness = roughness
        self.reflectivity = reflectivity
        self.emissionStrength = emissionStrength

    def toDict(self):
        return {"debugColor": [*self.debugColor], "color": [*self.color], "roughness": self.roughness,
                "reflectivity": self.reflectivity, "emissionStrength": self.emissionStrength}

    @staticmethod
    def fromDict(data):
        return Shader(debugColor=RGB(*data.get("debugColor", [0, 0, 0])), color=RGB(*data.get("color", [0, 0, 0])),
                      roughness=data.get("roughness", 1), reflectivity=data.get("reflectivity", 1),
                      emissionStrength=data.get("emissionStrength", 0))

    def __repr__(self):
        return (f"debugColor: {self.debugColor}\ncolor: {self.color}\nroughness: {self.roughness}\n"
                f"reflectivity: {self.reflectivity}\nemissionStrength: {self.emissionStrength}\n")


class Material:
    """Volume properties: index of refraction (T3DE.py:730-745)."""

    def __init__(self, ior=1):
        self.ior = ior

    def toDict(self):
        return {"ior": self.ior}

    @staticmethod
    def fromDict(data):
        return Material(ior=data.get("ior", 1))


def _read_stl_triangles(filename):
    """-> list of 3 corner tuples per facet, binary or ASCII STL, file order."""
    with open(filename, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        (count,) = _struct.unpack_from("<I", data, 80)
        if 84 + 50 * count == len(data):
            tris = []
            for i in range(count):
                vals = _struct.unpack_from("<12f", data, 84 + 50 * i)
                tris.append((vals[3:6], vals[6:9], vals[9:12]))
            return tris
    corners = [tuple(float(v) for v in line.split()[1:4])
               for line in data.decode("ascii", "replace").splitlines() if line.strip().startswith("vertex")]
    if not corners and data.strip() and b"facet" not in data and b"endsolid" not in data:
        # neither the size a binary file's facet count implies nor a single ASCII vertex: a cut-off or foreign file
        raise ValueError(f"{filename}: not a complete binary STL file (size {len(data)} does not fit its facet count) "
                         "and not an ASCII one")
    if len(corners) % 3:
        raise ValueError(f"{filename}: ASCII STL with {len(corners)} vertex lines, not a multiple of three")
    return [tuple(corners[i:i + 3]) for i in range(0, len(corners), 3)]


class Mesh:
    """Indexed triangle mesh: ``vertices`` (Vec3) and ``faces`` (index triples) (T3DE.py:749-821)."""

    def __init__(self, vertices, faces):
        self.vertices = vertices
        self.faces = faces

    def flipNormals(self):
        self.faces[:] = [face[::-1] for face in self.faces]

    def faceVertices(self, face):
        return [self.vertices[face[0]], self.vertices[face[1]], self.vertices[face[2]]]

    @staticmethod
    def centroid(v0, v1, v2):
        return Vec3((v0.x + v1.x + v2.x) / 3, (v0.y + v1.y + v2.y) / 3, (v0.z + v1.z + v2.z) / 3)

    @staticmethod
    def fromSTL(filename):
        """STL import with the reference's conventions (T3DE.py:778-790): y and z are exchanged so
        that y is up, and every face's winding is reversed.  Identical corners are merged in
        first-occurrence order (the reference delegates merging to trimesh; only the per-face
        coordinates and the face order influence rendering, and those are kept)."""
        index, vertices, faces = {}, [], []
        for tri in _read_stl_triangles(filename):
            ids = []
            for corner in tri:
                if corner not in index:
                    index[corner] = len(vertices)
                    vertices.append(Vec3(float(corner[0]), float(corner[2]), float(corner[1])))
                ids.append(index[corner])
            faces.append(tuple(reversed(ids)))
        return Mesh(vertices, faces)

    def toDict(self):
        return {"vertices": [[*v] for v in self.vertices], "faces": [[int(i) for i in f] for f in self.faces]}

    @staticmethod
    def fromDict(data):
        return Mesh([Vec3(*v) for v in data.get("vertices", [])],
                    [tuple(int(i) for i in f) for f in data.get("faces", [])])

    def __repr__(self):
        return "\n".join(map(str, self.vertices)) + "\n\n" + "\n".join(map(str, self.faces)) + "\n"


class BVHNode:
    """Axis-aligned box hierarchy over a mesh's faces (T3DE.py:880-949)."""

    def __init__(self, box=(None, None), left=None, right=None, faces=None):
        self.box = box
        self.left = left
        self.right = right
        self.faces = faces

    def merge(self, other):
        (alo, ahi), (blo, bhi) = self.box, other.box
        return (Vec3(min(alo.x, blo.x), min(alo.y, blo.y), min(alo.z, blo.z)),
                Vec3(max(ahi.x, bhi.x), max(ahi.y, bhi.y), max(ahi.z, bhi.z)))

    def toDict(self):
        return {"box": [None if v is None else [*v] for v in self.box],
                "left": self.left.toDict() if self.left else None,
                "right": self.right.toDict() if self.right else None,
                "faces": None if self.faces is None else [[int(i) for i in f] for f in self.faces]}

    @staticmethod
    def fromDict(data):
        try:
            left, right = data.get("left"), data.get("right")
            left = BVHNode.fromDict(left) if left is not None else None
            right = BVHNode.fromDict(right) if right is not None else None
        except RecursionError:
            left, right = BVHNode(), BVHNode()
        return BVHNode(box=[None if v is None else Vec3(*v) for v in data.get("box", [None, None])],
                       left=left, right=right, faces=data.get("faces"))

    def __repr__(self):
        shown = self.faces[:4] if self.faces else "not a leaf"
        return f"[{self.box[0]}, {self.box[1]}], {self.left}, {self.right}, {shown}\n"


_BOX_PAD = 0.06
_POINT_TYPES = ("light", "sphere", "camera", "empty")


class Scene:
    """Ordered collection of objects plus the bound camera (T3DE.py:608-690)."""

    def __init__(self):
        self.objects = []
        self.camera = None

    def getObject(self, id):
        return next((o for o in self.objects if o.id == id), None)

    def toDict(self):
        return {"camera": self.camera.toDict(), "objects": [o.toDict() for o in self.objects]}

    @staticmethod
    def fromDict(data):
        scene = Scene()
        cam_data = data.get("camera", {})
        camera = Camera.fromDict(cam_data)
        camera.scene = scene
        camera.id = cam_data.get("id", 0)
        scene.camera = camera
        scene.objects.append(camera)  # the bound camera always comes first
        for entry in data.get("objects", []):
            if entry.get("boundCam", False):
                continue
            obj = Object.fromDict(entry)
            scene.objects.append(obj)
            obj.scene = scene
            obj.id = entry.get("id", len(scene.objects))
        return scene

    def saveJSON(self, filepath):
        for obj in self.objects:
            obj.buildBVH()
        with open(filepath, "w") as f:
            _json.dump(self.toDict(), f)

    @staticmethod
    def fromJSON(filepath):
        with open(filepath, "r") as f:
            return Scene.fromDict(_json.load(f))


class Object:
    """Anything placed in a scene; ``type`` is 'mesh', 'sphere', 'light', 'camera' or 'empty'
    (T3DE.py:952-1211)."""

    def __init__(self, scene, name=None, mesh=None, position=None):
        self.__scene = scene
        self.id = max([-1] + [o.id for o in scene.objects]) + 1
        self.name = "Empty" if name is None else name
        self.position = Vec3(0, 0, 0) if position is None else position
        self.rotation = Euler(0, 0, 0)
        self.scale = Scale(1, 1, 1)
        self.type = "empty" if mesh is None else "mesh"
        self.mesh = _copy.deepcopy(mesh)
        self.material = Material()
        self.bvh = BVHNode()
        scene.objects.append(self)
        self._defaultShaders()

    @property
    def scene(self):
        return self.__scene

    @scene.setter
    def scene(self, scene):
        self.__scene = scene
        self.id = len(scene.objects)

    def _defaultShaders(self):
        self.shaders = [Shader()]
        if self.mesh is not None:
            self.shaderIndices = [0] * len(self.mesh.faces)
        elif self.type in _POINT_TYPES:
            self.shaderIndices = [0]
        else:
            self.shaderIndices = []

    # -- shaders ---------------------------------------------------------------------------
    def setShader(self, shader, faces=None):
        if faces is None:
            if self.type in _POINT_TYPES:
                self.shaders[0] = shader
                self.shaderIndices = [0]
                return
            faces = range(len(self.mesh.faces))
        if shader not in self.shaders:
            self.shaders.append(shader)
        slot = self.shaders.index(shader)
        for i in faces:
            self.shaderIndices[i] = slot

    def getShader(self, face=None):
        """Shader of a face.  Lights and cameras answer None: the reference's type list contains
        the single string 'light, camera' (T3DE.py:1052), and callers rely on the None."""
        if self.type == "mesh":
            return self.shaders[self.shaderIndices[self.mesh.faces.index(face)]]
        if self.type in ("empty", "sphere"):
            return self.shaders[0]
        return None

    def loadShadersFromJSON(self, filepath):
        with open(filepath, "r") as f:
            data = _json.load(f)
        shaders, faces = data.get("shaders", {}), data.get("faces", {})
        if len(faces) > len(self.mesh.faces):
            faces = faces[:len(self.mesh.faces)]
        self.shaders = list(shaders.values())
        self.shaderIndices = list(faces.values())

    # -- geometry --------------------------------------------------------------------------
    def setTransforms(self):
        """Bake position/rotation/scale into the mesh vertices and reset them (T3DE.py:1057-1071)."""
        if self.type != "mesh":
            return
        self.mesh.vertices = [v.applyTransforms(self.position, self.scale, self.rotation)
                              for v in self.mesh.vertices]
        self.position, self.scale, self.rotation = Vec3(0, 0, 0), Scale(1, 1, 1), Euler(0, 0, 0)

    def boundingBox(self, faces=None):
        if self.type == "mesh":
            faces = self.mesh.faces if faces is None else faces
            pts = [self.mesh.vertices[i].applyTransforms(self.position, self.scale, self.rotation)
                   for face in faces for i in face]
            lo = Vec3(*(min(p[a] for p in pts) - _BOX_PAD for a in range(3)))
            hi = Vec3(*(max(p[a] for p in pts) + _BOX_PAD for a in range(3)))
            return lo, hi
        if self.type in ("sphere", "light"):
            r = self.scale.x / 2
            half, pad = Vec3(r, r, r), Vec3(_BOX_PAD, _BOX_PAD, _BOX_PAD)
            return self.position - half - pad, self.position + half + pad
        if self.type in ("empty", "camera"):
            return self.position, self.position
        return None

    def buildBVH(self, faces=None, depth=0, maxTrianglesPerLeaf=8, maxDepth=100):
        """Median split on the centroid along axis depth%3, at most 8 faces per leaf
        (T3DE.py:985-1022)."""
        if self.type != "mesh":
            return BVHNode(box=self.boundingBox())
        faces = list(self.mesh.faces) if faces is None else faces
        box = self.boundingBox(faces)
        thin = any(abs(box[0][a] - box[1][a]) <= 0.0006 for a in range(3))
        if len(faces) <= maxTrianglesPerLeaf or thin or depth >= maxDepth:
            return BVHNode(box=box, faces=faces)
        axis = depth % 3
        verts = self.mesh.vertices
        faces.sort(key=lambda f: (verts[f[0]][axis] + verts[f[1]][axis] + verts[f[2]][axis]) / 3)
        half = len(faces) // 2
        lo = self.buildBVH(faces[:half], depth + 1, maxTrianglesPerLeaf, maxDepth)
        hi = self.buildBVH(faces[half:], depth + 1, maxTrianglesPerLeaf, maxDepth)
        return BVHNode(box=lo.merge(hi), left=lo, right=hi)

    # -- (de)serialisation -----------------------------------------------------------------
    def toDict(self):
        data = {"id": self.id, "name": self.name, "type": self.type,
                "position": [*self.position], "rotation": [*self.rotation], "scale": [*self.scale],
                "shaders": [s.toDict() for s in self.shaders], "shaderIndices": self.shaderIndices,
                "material": self.material.toDict(), "bvh": self.bvh.toDict()}
        if self.type == "mesh":
            data["mesh"] = self.mesh.toDict()
        elif self.type == "camera":
            data["length"] = self.length
            data["boundCam"] = self is self.scene.camera
        elif self.type == "light":
            data["strength"] = self.strength
        return data

    @staticmethod
    def fromDict(data):
        """Build a detached object (it lives in a private empty Scene until re-parented)."""
        kind = data.get("type", "empty")
        if kind == "mesh":
            obj = Object(scene=Scene())
            obj.mesh = Mesh.fromDict(data.get("mesh", {"vertices": [], "faces": []}))
        elif kind == "camera":
            obj = Camera(scene=Scene())
            obj.length = data.get("length", 1)
        elif kind == "light":
            obj = Light(scene=Scene())
            obj.strength = data.get("strength", 1)
        elif kind == "sphere":
            obj = Sphere(scene=Scene())
        else:
            obj = Object(scene=Scene())
        obj.name = data.get("name", "Empty Object")
        obj.type = kind
        obj.position = Vec3(*data.get("position", [0, 0, 0]))
        obj.rotation = Euler(*data.get("rotation", [0, 0, 0]))
        obj.scale = Scale(*data.get("scale", [0, 0, 0]))
        obj.shaders = [Shader.fromDict(s) for s in data.get("shaders", [])]
        obj.shaderIndices = data.get("shaderIndices", [])
        obj.material = Material.fromDict(data.get("material", {}))
        obj.bvh = BVHNode.fromDict(data.get("bvh", {}))
        return obj


_CUBE_CORNERS = [(1, 1, 1), (-1, 1, 1), (-1, -1, 1), (1, -1, 1), (1, 1, -1), (-1, 1, -1), (-1, -1, -1), (1, -1, -1)]
_CUBE_FACES = [(0, 1, 2), (2, 3, 0), (0, 3, 7), (7, 4, 0), (6, 2, 1), (1, 5, 6),
               (5, 1, 0), (0, 4, 5), (3, 2, 6), (6, 7, 3), (6, 5, 4), (4, 7, 6)]


class Cube(Object):
    """Mesh object preset: the cube [-1, 1]^3 as 12 triangles (T3DE.py:1215-1244)."""

    def __init__(self, scene, name="Cube", position=None):
        super().__init__(scene, name=name, position=position)
        self.type = "mesh"
        self.mesh = Mesh([Vec3(*c) for c in _CUBE_CORNERS], list(_CUBE_FACES))
        self.bvh = self.buildBVH()
        self._defaultShaders()


class Sphere(Object):
    """Analytic sphere of diameter ``scale.x`` (T3DE.py:1248-1253)."""

    def __init__(self, scene, name="Sphere", position=None):
        super().__init__(scene, name=name, position=position)
        self.type = "sphere"
        self.bvh = self.buildBVH()
        self._defaultShaders()


class Camera(Object):
    """Pinhole camera; ``length`` is the focal length, the image plane sits at length/20
    (T3DE.py:1257-1272).  The first camera created in a scene becomes its bound camera."""

    def __init__(self, scene, name="Camera", length=20, position=None):
        super().__init__(scene, name=name, position=position)
        self.__unit_length = length / 20
        self.type = "camera"
        if scene.camera is None:
            scene.camera = self
        self.bvh = self.buildBVH()
        self._defaultShaders()

    @property
    def length(self):
        return self.__unit_length * 20

    @length.setter
    def length(self, value):
        self.__unit_length = value / 20


class Light(Object):
    """Point light drawn as a sphere of diameter ``scale.x`` (T3DE.py:1276-1282)."""

    def __init__(self, scene, name="Light", position=None, strength=1):
        super().__init__(scene, name=name, position=position)
        self.type = "light"
        self.strength = strength
        self.bvh = self.buildBVH()
        self._defaultShaders()
