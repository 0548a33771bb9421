"""RayCastCollision throughput in config 4's world (full-body robot + 4096-hull soup): N rays against one
configuration through the C-ABI (tohc_ray_cast), and the CPU checker on a bounded sample beside it.
  python scripts/probe_rays.py [n_rays] [--no-cpu]      -> one JSON line
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import collision_checker as CC, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
s, rb, static_T, d = synth.config_fullbody_soup()
cc = CC.CreateCollisionChecker(s, static_T)
cc.SetRobot(rb, ignore_zero_state=False)
cc.SetContactDistance(d)
rng = np.random.default_rng(7)
q = synth.fullbody_trajectory(rb, rng, 100)[37]
cc.SetDOFValues(q)
a = np.column_stack([rng.uniform(-0.5, 4.0, n), rng.uniform(-1.8, 1.8, n), rng.uniform(1.5, 2.5, n)])
b = np.column_stack([a[:, 0] + rng.uniform(-0.5, 0.5, n), a[:, 1] + rng.uniform(-0.5, 0.5, n), rng.uniform(-1.0, 0.2, n)])
for _ in range(2):
    fr = cc.RayCastCollisions(a, b)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    fr = cc.RayCastCollisions(a, b)
dt = (time.perf_counter() - t0) / reps
out = {"workload": "config 4 world: %d rays x %d convex children, one configuration" % (n, len(s.arrays["child_shape"])), "rays": n,
       "hits": int((fr < 1).sum()), "ms_per_call_host_delivered": dt * 1e3, "rays_per_s": n / dt, "ray_child_tests_per_s": n * len(s.arrays["child_shape"]) / dt}
if "--no-cpu" not in sys.argv:
    from oracle import bindings as B
    fk = cc.eng.forward_kinematics(q[None])[0]
    objT = static_T.copy()
    for i, o in enumerate(rb.link_obj):
        if o >= 0:
            objT[o] = fk[i]
    kind = "reference" if B.ref_available() else "port"
    w = (B.ref_world if kind == "reference" else B.oracle_world)(s, cap=16)
    w.set_contact_distance(d)
    m = min(n, 2000)
    t0 = time.perf_counter()
    fo = w.ray_cast(objT, a[:m], b[:m])
    dc = time.perf_counter() - t0
    out["cpu_baseline"] = {"kind": kind, "cores": 1, "rays_per_s": m / dc, "sample": "%d rays of the same batch" % m,
                           "agrees": bool(np.array_equal(fo.view(np.uint32), fr[:m].view(np.uint32)))}
print(json.dumps(out))
