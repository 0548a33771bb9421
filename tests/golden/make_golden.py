"""Generates tests/golden/*.npz from the REFERENCE's own vendored Bullet (oracle/_ref/libref.so, built from
/root/reference by oracle/Makefile).  Run in the build container only:  python tests/golden/make_golden.py

Each fixture holds the flat scene arrays, the poses, the contact distance and the contacts Bullet reported
(canonical order).  Fixtures are data: inputs + expected outputs; no reference source is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bindings as B  # noqa: E402
from trajopt_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def scene_arrays(s):
    return {"scene_" + k: v for k, v in s.arrays.items()}


def main():
    assert B.ref_available(), "oracle/_ref/libref.so missing: make -C oracle ref"
    cases = []
    for seed, spread in [(1, 0.6), (2, 0.3), (3, 0.3), (4, 0.45), (5, 0.25), (6, 0.6)]:
        rng = np.random.default_rng(seed)
        s, objT, robot = synth.random_scene(rng, spread=spread)
        T1 = objT[robot].copy()
        T1[:, 4:] += rng.normal(size=(len(robot), 3)) * 0.15
        T1[:, :4] += rng.normal(size=(len(robot), 4)) * 0.2
        T1[:, :4] /= np.linalg.norm(T1[:, :4], axis=1, keepdims=True)
        w = B.ref_world(s)
        out = dict(scene_arrays(s), objT=objT, robot=np.array(robot, np.int32), T1=T1, dists=np.array([0.0, 0.05, 0.2], np.float32))
        for i, d in enumerate(out["dists"]):
            w.set_contact_distance(float(d))
            out["disc_%d" % i] = B.sort_contacts(w.links_vs_all(objT, robot))
            out["cast_%d" % i] = B.sort_contacts(w.cast_vs_all(objT, robot, objT[robot], T1))
        w.close()
        np.savez_compressed(os.path.join(HERE, "random_scene_%d.npz" % seed), **out)
        cases.append((seed, sum(len(out["disc_%d" % i]) for i in range(3)), sum(len(out["cast_%d" % i]) for i in range(3))))
    print("wrote", cases)


if __name__ == "__main__":
    main()
