#!/usr/bin/env python3
"""Counterpart of the reference's demo script: build the Cornell scene through the T3DE API, write it to
``cornellScene.json`` and render it with ``T3DR.Photon`` (cornellTest.py:7-94 builds the scene, :107-130 the renderer,
:134-137 the JSON, :139-155 the render and the image file named ``<name>|<seconds>|<rays>.png`` -- commented out
there, since the render takes hours in Python).

    python cornell_test.py [--width 1440 --height 900 --bounces 3 --samples 2 --out renders] [--light]

Needs an MI355X (the tracing runs in libt3d_hip.so; there is no CPU rendering path).  Writes
``<out>/cornellScene.json`` and ``<out>/cornell|<time>|<rays>.png``.
"""
import argparse
import json
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def write_png(path, rgb):
    """A minimal 8-bit RGB PNG writer (the reference uses cv2.imwrite, which is not a dependency here)."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1440)      # the reference's default image size (T3DR.py:519-525)
    ap.add_argument("--height", type=int, default=900)
    ap.add_argument("--bounces", type=int, default=3)       # cornellTest.py:108-109
    ap.add_argument("--samples", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--light", action="store_true", help="add the Light object of the benchmark scene (the demo script has none)")
    ap.add_argument("--out", default="renders")
    args = ap.parse_args()

    from T3DE import RGB
    from T3DR import Photon
    from raytracing_amd.workloads import cornell_scene

    scene = cornell_scene(with_light=args.light)
    renderer = Photon(scene=scene, options={
        "bounces": args.bounces, "samples": args.samples, "aabb": True, "bvh": True, "wait": False, "debug": False,
        "lights": True, "emissionSampleDensity": 1, "ambient": RGB(32, 32, 32), "threads": 1,
        "width": args.width, "height": args.height, "seed": args.seed})
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, "cornellScene.json"), "w") as f:
        json.dump(scene.toDict(), f)
    render = renderer.render()
    name = os.path.join(args.out, f"cornell|{render.time:.4f}|{render.rays}.png")
    write_png(name, render.imageAs("np"))
    print(f"{name}: {render.width}x{render.height}, {render.rays} rays in {render.time * 1e3:.2f} ms "
          f"({render.stats['queries']} closest-hit queries)")
    return render


if __name__ == "__main__":
    main()
