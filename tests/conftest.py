import json
import os
import sys

import numpy as np
import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(g["meta"]))
    return g, meta


def flat_for(scene_file):
    from raytracing_amd.flatten import flatten_scene
    from raytracing_amd.scene import Scene
    return flatten_scene(Scene.fromJSON(os.path.join(SCENES, scene_file)))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_ctx():
    from raytracing_amd import _native
    ctx = _native.Context(0)
    yield ctx
    ctx.close()
