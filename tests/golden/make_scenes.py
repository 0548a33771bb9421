"""Generates tests/golden/ref_scenes.json from the reference's own scene files (run in the build container, where
/root/reference exists; the GPU box only sees the committed JSON).

    python tests/golden/make_scenes.py

data/{tabletop,bookshelves,kitchen_counter,kitchen}.env.xml are the obstacle worlds of python_examples/benchmark.py:3
and fullbody_plan.py:56 (BASELINE.json configs[2]): one static KinBody per box, each box a Geom with its own
translation / quaternion, i.e. a geometry that is NOT at the link's identity -- the reference wraps such a link in a
btCompoundShape (bullet_collision_checker.cpp:224-250).  The fixture keeps only the numbers: name, half extents,
translation, quaternion (w x y z).
"""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def boxes(path):
    out = []
    for kb in ET.parse(path).getroot().iter("KinBody"):
        for g in kb.iter("Geom"):
            assert g.get("type") == "box"
            num = lambda tag: [float(x) for x in g.find(tag).text.split()]
            out.append(dict(name=kb.get("name"), extents=num("extents"), translation=num("translation"), quat=num("quat")))
    return out


if __name__ == "__main__":
    scenes = {n: boxes(os.path.join(REF, n + ".env.xml")) for n in ("tabletop", "bookshelves", "kitchen_counter", "kitchen")}
    with open(os.path.join(HERE, "ref_scenes.json"), "w") as f:
        json.dump(scenes, f, indent=0)
    print({k: len(v) for k, v in scenes.items()})
