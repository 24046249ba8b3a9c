"""Drop-in module name for scripts written against the reference engine: ``from T3DE import Scene, ...``.
The implementation lives in ``raytracing_amd.scene``."""
from raytracing_amd.scene import *  # noqa: F401,F403
from raytracing_amd.scene import (RGB, BVHNode, Camera, Cube, Euler, Light, Material, MathematicalPlane,  # noqa: F401
                                  MathematicalSphere, Matrix9, Mesh, Object, Ray, Scale, Scene, Shader, Sphere,
                                  TraceTreeNode, Vec3)
