"""MI355X-native implementation of the T3D ``Photon`` ray-tracing path.

``raytracing_amd.scene``   scene model (the ``T3DE`` API)
``raytracing_amd.photon``  ``Photon`` / ``Rendered`` (the ``T3DR`` API), backed by libt3d_hip.so
``raytracing_amd.flatten`` scene -> flat arrays for the C ABI (include/t3d.h)
``raytracing_amd.tiles``   row-tile partitioner + gather for one-process-per-GPU rendering
"""
from .scene import (RGB, BVHNode, Camera, Cube, Euler, Light, Material, MathematicalPlane, MathematicalSphere,
                    Matrix9, Mesh, Object, Ray, Scale, Scene, Shader, Sphere, TraceTreeNode, Vec3)

__all__ = ["RGB", "BVHNode", "Camera", "Cube", "Euler", "Light", "Material", "MathematicalPlane",
           "MathematicalSphere", "Matrix9", "Mesh", "Object", "Ray", "Scale", "Scene", "Shader", "Sphere",
           "TraceTreeNode", "Vec3"]
