"""Generates tests/golden/shape_hull.npz from the REFERENCE's own vendored Bullet (oracle/_ref/libref.so, ref_shape_hull):
btShapeHull over a btConvexTriangleMeshShape as createShapePrimitive builds it for a GT_TriMesh geometry
(bullet_collision_checker.cpp:174-203).  Run in the build container only:  python tests/golden/make_shape_hull.py

Per mesh: vertices (float32), triangles (int32) and the points the reference's btConvexHullShape ends up with
(float32, in its order).  Data only.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings as B  # noqa: E402
from trajopt_amd import synth  # noqa: E402

out = {}
sizes = [20, 49, 50, 51, 64, 120, 400, 1500]
k = 0
for kind in range(4):
    for n in sizes:
        rng = np.random.default_rng(7000 + k)
        v, t = synth.random_trimesh(rng, n, kind)
        out["v%d" % k], out["t%d" % k], out["h%d" % k] = v, t, B.ref_shape_hull(v, t)
        k += 1
out["n"] = np.int32(k)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "shape_hull.npz"), **out)
print("wrote %d meshes; hull sizes %s" % (k, [len(out["h%d" % i]) for i in range(k)]))
