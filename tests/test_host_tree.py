"""The host-built hierarchy (raytracing_amd/csrc/t3d_host_tree.h) is a tree in the device's numbering.

The checker (tools/host_tree_check.cpp) builds it over generated boxes -- a uniform soup, a dense mesh inside a few
large walls, coincident boxes, runs of duplicates, sizes in geometric progression -- and walks it from the root:
every primitive at exactly one leaf position, child ranges nested and adjacent, node ids at an end of their range
(what k_lbvh_hierarchy produces and refit / packing rely on), parents consistent.  No GPU involved.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("host_tree") / "host_tree_check")
    subprocess.run([cxx, "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tools", "host_tree_check.cpp")], check=True)
    return exe


def _run(checker, *sizes, threads=None):
    env = dict(os.environ)
    if threads is not None:
        env["T3D_HOST_TREE_THREADS"] = str(threads)
    res = subprocess.run([checker, *map(str, sizes)], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = [line.split() for line in res.stdout.splitlines()]
    assert rows and all(r[0] == "ok" for r in rows)
    return res.stdout.splitlines()


def test_small_and_degenerate_inputs(checker):
    lines = _run(checker, 0, 1, 2, 3, 4, 5, 7, 8, 9, 33, 64)
    assert len(lines) == 11 * 5


def test_thousands_of_boxes(checker):
    lines = _run(checker, 1000, 20000)
    by = {(l.split(" n=")[0][3:].strip(), int(l.split(" n=")[1].split()[0])): l for l in lines}
    depth = lambda l: int(l.split("depth")[1].split()[0])
    cost = lambda l: float(l.split("cost")[1].split()[0])
    # a soup of 20 000 is cut evenly (depth near log2 n); sizes in geometric progression peel off box after box
    assert depth(by[("soup", 20000)]) <= 24
    assert depth(by[("graded sizes", 20000)]) > 48          # what the library's depth check is for
    # a dense mesh inside large walls: the walls are split off at the top, so a ray that misses the mesh pays for few nodes
    assert cost(by[("room+mesh", 20000)]) < 0.3 * cost(by[("soup", 20000)])


def test_large_inputs_are_cut_on_several_threads_with_the_same_result(checker):
    """From 32 768 boxes on the pieces below the top of the tree are cut on several threads (and the items of the top nodes
    binned in slices): the tree must not depend on how many."""
    one = _run(checker, 70000, threads=1)
    four = _run(checker, 70000, threads=4)
    key = lambda lines: [(l.split(" n=")[0], l.split("cost")[1]) for l in lines]      # cost, depth and a hash of the whole tree
    assert key(one) == key(four)
