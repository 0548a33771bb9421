"""Generates tests/golden/rays.npz from the REFERENCE's own vendored Bullet (oracle/_ref/libref.so): RayCastCollision
(bullet_collision_checker.cpp:352-360 -> btCollisionWorld::rayTest) on the scenes of random_scene_*.npz.
Run in the build container only:  python tests/golden/make_rays.py

Per scene: ray end points (float64, as the caller hands them over) and the ClosestRayResultCallback's
m_closestHitFraction after rayTest (float32; 1 = no hit) at contact distances 0 and 0.05.  Data only.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings as B  # noqa: E402
from trajopt_amd import model as M  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rays_for(rng, objT, n=300):
    ext = float(np.abs(objT[:, 4:7]).max()) * 1.5 + 0.2
    a = rng.uniform(-ext, ext, (n, 3))
    b = rng.uniform(-ext, ext, (n, 3))
    k = n // 10
    a[:k] = objT[rng.integers(0, len(objT), k), 4:7]           # starting at an object's origin (usually inside it)
    b[k:2 * k] = objT[rng.integers(0, len(objT), k), 4:7]      # aimed at an object's origin
    b[2 * k:2 * k + 5] = a[2 * k:2 * k + 5]                    # zero length
    return a, b


def main():
    assert B.ref_available(), "oracle/_ref/libref.so missing: make -C oracle ref"
    out, stats = {}, []
    for path in sorted(glob.glob(os.path.join(HERE, "random_scene_*.npz"))):
        z = np.load(path, allow_pickle=False)
        tag = os.path.basename(path)[len("random_scene_"):-4]
        s = M.scene_from_arrays({k[6:]: z[k] for k in z.files if k.startswith("scene_")})
        rng = np.random.default_rng(1000 + int(tag))
        a, b = rays_for(rng, z["objT"])
        w = B.ref_world(s)
        out["from_" + tag], out["to_" + tag] = a, b
        for i, d in enumerate((0.0, 0.05)):
            w.set_contact_distance(d)
            out["frac_%s_%d" % (tag, i)] = w.ray_cast(z["objT"], a, b)
        w.close()
        stats.append((tag, int((out["frac_%s_0" % tag] < 1).sum()), len(a)))
    np.savez_compressed(os.path.join(HERE, "rays.npz"), **out)
    print("wrote (scene, hits, rays):", stats)


if __name__ == "__main__":
    main()
