"""Generates tests/golden/ref_problems.json from the reference's own problem files (run in the build container, where
/root/reference exists; the GPU box only sees the committed JSON).

    python tests/golden/make_problems.py

BASELINE.json configs[0] / configs[1]: data/arm_around_table.json (n_steps 10, init trajectory :33-42) and
data/arm_around_table_continuous.json (n_steps 6, init trajectory :36-41), both in the world of data/table.xml.  The
fixture keeps only the numbers a collision term needs: n_steps, the collision cost's coeffs / dist_pen / continuous flag
(a missing "continuous" key means TRUE, problem_description.cpp:493 -- so arm_around_table.json is a cast problem as
shipped), the contact distance the reference derives from them (max(dist_pen) + 0.04, problem_description.cpp:548-549),
the init trajectory and the joint target.
"""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def problem(path):
    with open(path) as f:
        j = json.load(f)
    coll = [c for c in j["costs"] if c["type"] == "collision"]
    assert len(coll) == 1
    p = coll[0]["params"]
    traj = j["init_info"]["data"]
    assert j["init_info"]["type"] == "given_traj" and len(traj) == j["basic_info"]["n_steps"]
    return dict(n_steps=j["basic_info"]["n_steps"], manip=j["basic_info"]["manip"], continuous=bool(p.get("continuous", True)),
                coeffs=p["coeffs"], dist_pen=p["dist_pen"], contact_distance=max(p["dist_pen"]) + 0.04,
                joint_target=[c for c in j["constraints"] if c["type"] == "joint"][0]["params"]["vals"], init_traj=traj)


def table(path):
    g = ET.parse(path).getroot().find(".//Geom")
    assert g.get("type") == "box"
    num = lambda tag: [float(x) for x in g.find(tag).text.split()]
    return dict(extents=num("extents"), translation=num("Translation"))


if __name__ == "__main__":
    out = {"arm_around_table": problem(os.path.join(REF, "arm_around_table.json")),
           "arm_around_table_continuous": problem(os.path.join(REF, "arm_around_table_continuous.json")),
           "table": table(os.path.join(REF, "table.xml"))}
    with open(os.path.join(HERE, "ref_problems.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (v.get("n_steps"), v.get("continuous")) for k, v in out.items()})
