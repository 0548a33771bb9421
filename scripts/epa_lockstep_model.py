"""CPU model of k_tail_epa's lock step (no GPU): the EPA run lengths of config 4 from the host emulation (EMU_EPA_LOG: one
line per run with its iteration count and the pair's child numbers), then what a wavefront of eight runs costs under
different orders of the queue and under two-phase schemes.  A wavefront costs max(iterations of its 8 runs) + 2.5 (the
shared prologue / epilogue, in iterations).  It is where filing the queue by the class of the pair's GJK2 came from.

  python scripts/epa_lockstep_model.py [problems]      (default 6; ~1 minute)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LOG = "/tmp/epa_log.txt"
os.environ["EMU_EPA_LOG"] = LOG
if os.path.exists(LOG):
    os.remove(LOG)
from oracle import bindings as B          # noqa: E402
from tests import util                    # noqa: E402
from trajopt_amd import model as M, synth  # noqa: E402

nP = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s, rb, static_T, d = synth.config_fullbody_soup(npts=0)
w = util.emu_world(s)
w.set_contact_distance(d)
link_obj = np.array(rb.link_obj)
robot_objs = [int(o) for o in link_obj if o >= 0]


def poses(model, q):
    fk = B.oracle_fk(model, q)
    T = np.array(static_T, dtype=np.float64).copy()
    for i, o in enumerate(link_obj):
        if o >= 0:
            T[o] = fk[i]
    return T


# zero-state self-collision exclusions (collision_checker.cpp:41-56), as the bench's checker sets them
zero = M.RobotModel.__new__(M.RobotModel)
zero.__dict__.update(rb.__dict__)
zero.q_robot = np.zeros(max(rb.n_dof, 1)); zero.base_T = (1, 0, 0, 0, 0, 0, 10); zero.joint_inds, zero.affine = list(range(rb.n_dof)), 0
zero.finalize()
c0 = w.links_vs_all(poses(zero, np.zeros(zero.K)), robot_objs)
allowed = np.ones((s.n_objects, s.n_objects), np.uint8)
allowed[c0["objA"], c0["objB"]] = 0
allowed[c0["objB"], c0["objA"]] = 0
w.set_allowed(allowed)

rows = []   # problem, t, cast, iterations, query child, other child, distance, passes of Penetration's GJK2, GJK iterations
for p in range(nP):
    traj = synth.fullbody_trajectory(rb, np.random.default_rng(1000 + p), 100)
    Ts = [poses(rb, q) for q in traj]
    for t in range(100):
        for cast in (0, 1):
            if cast and t == 99:
                continue
            pos = os.path.getsize(LOG) if os.path.exists(LOG) else 0
            if cast:
                w.cast_vs_all(Ts[t], robot_objs, Ts[t][robot_objs], Ts[t + 1][robot_objs])
            else:
                w.links_vs_all(Ts[t], robot_objs)
            if os.path.exists(LOG):
                with open(LOG) as f:
                    f.seek(pos)
                    for line in f:
                        a = line.split()
                        rows.append((p, t, cast, int(a[0]), int(a[7]), int(a[8]), float(a[5]), int(a[9]), int(a[11])))
r = np.array(rows)
rng = np.random.default_rng(0)


def cost(it, c0=2.5, G=8, cap=None):
    pad = (-len(it)) % G
    m = np.concatenate([it, np.zeros(pad)]).reshape(-1, G).max(axis=1)
    if cap is not None:
        m = np.minimum(m, cap)
    return (m + c0).sum()


it = r[:, 3]
bound = it.sum() / 8 + 2.5 * len(it) / 8
print("%d EPA runs, %.2f iterations on average, at most %d; histogram %s" % (len(it), it.mean(), it.max(), np.bincount(it.astype(int)).tolist()))
print("no-waste bound (every lane group busy all the time): %.0f" % bound)
g2 = r[:, 7]
cls = np.where(g2 <= 3, 0, np.where(g2 == 4, 1, np.where(g2 == 5, 2, 3)))   # k_tail_pen's classes
print("passes of Penetration's GJK2 against the EPA run that follows: correlation %.2f; by class (<= 3, 4, 5, >= 6 passes): share %s, mean EPA iterations %s"
      % (np.corrcoef(g2, it)[0, 1], [round(float((cls == c).mean()), 2) for c in range(4)], [round(float(it[cls == c].mean()), 1) if (cls == c).any() else None for c in range(4)]))
orders = {"as queued (random)": rng.permutation(len(r)),
          "FILED BY GJK2 CLASS (what k_tail_pen does)": np.lexsort((rng.uniform(size=len(r)), cls)),
          "filed by the exact number of GJK2 passes": np.lexsort((rng.uniform(size=len(r)), g2)),
          "sorted by (query child, other child, problem, t)": np.lexsort((r[:, 1], r[:, 0], r[:, 5], r[:, 4])),
          "sorted by penetration depth": np.argsort(r[:, 6]),
          "sorted by run length (what no predictor can know)": np.argsort(it)}
base = None
for k, o in orders.items():
    c = cost(it[o])
    base = base or c
    print("  %-52s %.0f = %.2f x bound, %.2f x as queued" % (k, c, c / bound, c / base))
sh = it[rng.permutation(len(it))]
for N in (6, 7, 8, 10):
    long_ = sh[sh > N]
    restart = cost(sh, cap=N) + cost(long_)
    pad = (-len(long_)) % 8
    resumed = cost(sh, cap=N) + ((np.concatenate([long_, np.full(pad, N)]).reshape(-1, 8).max(axis=1) - N) + 1.5).sum()
    print("  two phases, cut after %2d iterations: long runs started again %.2f x, resumed from saved pools %.2f x as queued" % (N, restart / base, resumed / base))
key = {(a[0], a[2], a[4], a[5], a[1]): a[3] for a in r}
pairs = [(v, key[(p, c, q, x, t + 1)]) for (p, c, q, x, t), v in key.items() if (p, c, q, x, t + 1) in key]
print("the same pair at t and t + 1: %d cases, correlation of run lengths %.2f" % (len(pairs), np.corrcoef(np.array(pairs).T)[0, 1]))


def refill_cost(itq, m, c0=2.5, G=8):
    """Lane groups that take their next run themselves: a refill (epilogue of the finished runs + prologue of the next:
    c0 iterations of wavefront time, the running groups masked off meanwhile) happens as soon as m groups are idle -- or
    all that are left; m = G is the lock step of today's rounds."""
    q = list(itq)
    qi = 0
    rem = np.zeros(G)
    t = 0.0
    while True:
        idle = int((rem <= 0).sum())
        if qi < len(q) and (idle >= m or idle == G):
            for g in range(G):
                if rem[g] <= 0 and qi < len(q):
                    rem[g] = q[qi]; qi += 1
            t += c0
            continue
        if idle == G:
            break
        if qi >= len(q):          # nothing left to hand out: run the rest down
            t += rem.max(); break
        step = np.sort(rem[rem > 0])[max(0, m - idle - 1)] if idle < m else 1   # until enough groups are idle
        rem = rem - step
        t += step
    return t


W = 2048 * 0 + 256   # wavefronts that share the queue in the model (each takes its runs in queue order, round robin)
for name in ("as queued (random)", "FILED BY GJK2 CLASS (what k_tail_pen does)"):
    o = orders[name]
    seq = it[o]
    for m in (8, 6, 4, 3, 2, 1):
        tot = sum(refill_cost(seq[wv::W], m) for wv in range(W))
        print("  self-refilling lane groups, refill when %d idle, %-44s %.0f = %.2f x bound" % (m, name + ":", tot, tot / bound))
