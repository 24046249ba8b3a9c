import json
import os
import sys

import numpy as np
import pytest


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")                  # scenes the fixture generator dumped
DATA = os.path.join(ROOT, "raytracing_amd", "data")      # the reference's own scene and mesh files


def scene_path(name):
    """A scene or mesh file by name: the reference's files ship with the package, generated ones sit with the fixtures."""
    p = os.path.join(DATA, name)
    return p if os.path.exists(p) else os.path.join(SCENES, name)


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
    return flatten_scene(Scene.fromJSON(scene_path(scene_file)))


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


def oracle_rows_threaded(oracle, flat, options, seed, rows, want_radiance=False, threads=None):
    """The oracle on the image rows ``rows`` (any list of row numbers), one row per call, the calls spread over host
    threads (ctypes releases the GIL inside the C library).  -> (ref dict with arrays stacked in the order of
    ``rows`` and summed counters, fragility array) -- what ``oracle.render`` and ``oracle.primary_fragility`` give for
    a regular row pattern, for an arbitrary one and in a fraction of the time."""
    import concurrent.futures as cf
    rows = list(rows)
    threads = threads or max(1, min(16, os.cpu_count() or 1, len(rows)))

    def one(r):
        ref = oracle.render(flat, options, seed=seed, row0=r, nrows=1, want_radiance=want_radiance)
        frag = oracle.primary_fragility(flat, dict(options), row0=r, nrows=1)
        return ref, frag

    with cf.ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(one, rows))
    ref = {k: np.concatenate([p[0][k] for p in parts]) for k in ("bgr", "prim", "prim_t")}
    ref["radiance"] = np.concatenate([p[0]["radiance"] for p in parts]) if want_radiance else None
    stats = {}
    for p in parts:
        for k, v in p[0]["stats"].items():
            if not isinstance(v, list):
                stats[k] = stats.get(k, 0) + v
    ref["stats"] = stats
    return ref, np.concatenate([p[1] for p in parts])
