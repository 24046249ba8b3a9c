#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of a HIP source compiled for gfx950 (no GPU needed):
    python tools/kernel_resources.py [file.hip]     default: raytracing_amd/csrc/t3d_hip.hip"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "raytracing_amd", "csrc", "t3d_hip.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize",
                           "-Wno-unused-value", "-S", "--cuda-device-only", "-I", os.path.join(root, "raytracing_amd", "csrc"),
                           "-I", os.path.join(root, "include"), "-o", out, src] + sys.argv[2:])
    text = open(out).read()
for block in text.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", block).group(1)
    print(f"{dem:32s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
