"""How many broadphase survivors of config 4 does the separating-axis pre-test remove? (host emulation, CPU only)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import util
from oracle import bindings as B
from trajopt_amd import synth, collision_checker as CC
s, rb, static_T, d = synth.config_fullbody_soup()
we = util.emu_world(s)
we.set_contact_distance(d)
# same exclusions as the bench: zero-state self collisions
wo = B.oracle_world(s); wo.set_contact_distance(0.1)
objs, links = rb.affected_links()
link_obj = np.array(rb.link_obj); have = link_obj >= 0
T0 = static_T.copy(); fk = B.oracle_fk(rb, np.zeros(rb.K)); T0[link_obj[have]] = fk[have]
allowed = np.ones((s.n_objects, s.n_objects), np.uint8)
robot_objs = set(int(o) for o in link_obj[have])
for c in wo.links_vs_all(T0, objs):
    if int(c['objA']) not in robot_objs or int(c['objB']) not in robot_objs:
        continue
    allowed[c['objA'], c['objB']] = allowed[c['objB'], c['objA']] = 0
we.set_allowed(allowed)
rng = np.random.default_rng(1000)
traj = synth.fullbody_trajectory(rb, rng, 100)
n = 0
for t in range(0, 99, 1):
    P = static_T.copy(); f0 = B.oracle_fk(rb, traj[t]); f1 = B.oracle_fk(rb, traj[t+1]); P[link_obj[have]] = f0[have]
    n += len(we.links_vs_all(P, objs))
    n += len(we.cast_vs_all(P, objs, f0[links], f1[links]))
lib = we.lib
lib.emu_pre.restype = C.c_long; lib.emu_iter_hist.restype = C.c_long
pre = [lib.emu_pre(i) for i in range(8)]
print("contacts", n, "pairs", sum(pre[:4]), "pass/axis1/axis2/axis3", pre[:4], "violations", pre[5:8])
print("iters", [lib.emu_iter_hist(i) for i in range(64)])
lib.emu_stat.restype = C.c_long
st = [lib.emu_stat(i) for i in range(6 + 72 + 12 + 1 + 130)]
print("gjk2 calls %d iters %d | epa calls %d iters %d faces(sum) %d" % tuple(st[:5]))
print("epa iteration histogram", st[6:6 + 72])
print("epa status", st[78:90], "gjk2 inside", st[90])
h = st[91:]
print("gjk2 iteration histogram (nonzero):", {i: v for i, v in enumerate(h) if v})
