"""Shared comparison helpers: contact sets are compared EXACTLY, values bit-for-bit unless a tolerance is given."""
import ctypes as C
import os

import numpy as np

from oracle import bindings as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("objA", "childA", "objB", "childB")
VALS = ("dist", "n", "ptA", "ptB", "ptB1", "time", "cctype")


def emu_world(scene):
    return B.World(C.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so")), "emu_", scene)


def contact_keys(c):
    return [tuple(int(c[k][i]) for k in KEYS) for i in range(len(c))]


def assert_same_contacts(a, b, tag="", tol=None):
    a, b = B.sort_contacts(a), B.sort_contacts(b)
    ka, kb = contact_keys(a), contact_keys(b)
    assert ka == kb, "%s: contact SETS differ: %d vs %d, symmetric difference %s" % (tag, len(ka), len(kb), sorted(set(ka) ^ set(kb))[:8])
    for f in VALS:
        if tol is None:
            assert np.array_equal(a[f], b[f]), "%s: field %s differs, max abs %g" % (
                tag, f, np.abs(a[f].astype(float) - b[f].astype(float)).max(initial=0))
        else:
            assert np.allclose(a[f], b[f], rtol=0, atol=tol), (tag, f)


def assert_same_all_vs_all(a, b, tag=""):
    """AllVsAll results: object pair in manifold order, distance, points and normal bit for bit.  (The reference's manifold
    points carry the compound algorithms' shape identifiers, which are stale for one side of a compound-compound pair, so
    child numbers are compared only between the engine and the oracle: pass them through when both sides have them.)"""
    def key(c):
        return sorted(zip(c["objA"].tolist(), c["objB"].tolist(), c["dist"].tolist(), map(tuple, c["ptA"].tolist()), map(tuple, c["ptB"].tolist()),
                          map(tuple, c["n"].tolist())))
    ka, kb = key(a), key(b)
    assert ka == kb, "%s: AllVsAll differs: %d vs %d contacts, e.g. %s" % (tag, len(ka), len(kb), sorted(set(ka) ^ set(kb))[:2])


def records_to_contacts(rec):
    """Engine records (RECORD_DTYPE) -> oracle CONTACT_DTYPE, plus (problem, t)."""
    c = np.zeros(len(rec), B.CONTACT_DTYPE)
    c["objA"], c["childA"], c["objB"], c["childB"] = rec["idA"], rec["childA"], rec["idB"], rec["childB"]
    for f in ("cctype", "time", "dist", "n", "ptA", "ptB", "ptB1"):
        c[f] = rec[f]
    return c


def lin_order(orec):
    c = orec["c"]
    return np.lexsort((c["childB"], c["objB"], c["childA"], c["objA"], orec["t"]))


def assert_same_linearization(rec, grads, orec, ograds, tag="", exact_values=True, jac_rtol=1e-5):
    """Engine (rec, grads) vs oracle (orec, ograds): same (t, A, B) set, same contact values, Jacobian rows within
    the north-star tolerance (1e-5 relative; FK runs on the device with the device's sin/cos)."""
    o = lin_order(orec)
    orec, ograds = orec[o], ograds[o]
    ke = list(zip(rec["t"].tolist(), rec["idA"].tolist(), rec["childA"].tolist(), rec["idB"].tolist(), rec["childB"].tolist()))
    ko = list(zip(orec["t"].tolist(), *[orec["c"][k].tolist() for k in KEYS]))
    assert ke == ko, "%s: linearization sets differ: %d vs %d; %s" % (tag, len(ke), len(ko), sorted(set(ke) ^ set(ko))[:8])
    c = records_to_contacts(rec)
    for f in VALS:
        if exact_values:
            assert np.array_equal(c[f], orec["c"][f]), "%s: field %s differs, max abs %g" % (
                tag, f, np.abs(c[f].astype(float) - orec["c"][f].astype(float)).max(initial=0))
        else:
            assert np.allclose(c[f], orec["c"][f], rtol=0, atol=1e-5), (tag, f)
    if grads is not None and len(rec):
        scale = max(1.0, float(np.abs(ograds).max()))
        assert np.abs(grads - ograds).max() <= jac_rtol * scale, "%s: gradient rows differ by %g" % (tag, np.abs(grads - ograds).max())
        k0 = np.abs(rec["const0"] - orec["const0"]).max()
        k1 = np.abs(rec["const1"] - orec["const1"]).max()
        assert max(k0, k1) <= jac_rtol * scale * 10, "%s: constants differ by %g" % (tag, max(k0, k1))


def write_flat_world(path, scene, rb, static_T, traj, d, dist_pen, coeff):
    """A scene + robot + trajectory as named arrays of numbers, the input of tests/cpp/host_unit (`linearize`).
    Values are printed with repr(), so doubles and fp32 values survive the text round trip bit for bit."""
    a, r = scene.arrays, rb.arrays
    items = {k: a[k] for k in ("kind", "param", "voff", "vcnt", "verts", "child_shape", "child_T", "obj_off", "obj_cnt", "obj_compound", "obj_group")}
    items.update({k: r[k] for k in ("parent", "jtype", "T_origin", "axis", "dof", "mult", "offs", "link_obj", "q_robot", "base_T", "joint_inds", "rot_axis")})
    items.update(static_T=static_T, traj=traj, d=[d], dist_pen=[dist_pen], coeff=[coeff], n_dof=[rb.n_dof], affine=[rb.affine],
                 n_joint_inds=[len(rb.joint_inds)])
    # GT_TriMesh geometries (Scene.add_trimesh): the mesh itself, so that the C++ checker does the btShapeHull reduction
    ns = len(a["kind"])
    mvoff, mvcnt, mtoff, mtcnt, mv, mt = np.zeros(ns), np.zeros(ns), np.zeros(ns), np.zeros(ns), [], []
    for sid, (v, t) in sorted(scene.mesh_src.items()):
        mvoff[sid], mvcnt[sid], mtoff[sid], mtcnt[sid] = sum(len(x) for x in mv), len(v), sum(len(x) for x in mt), len(t)
        mv.append(v); mt.append(t)
    items.update(mesh_voff=mvoff, mesh_vcnt=mvcnt, mesh_toff=mtoff, mesh_tcnt=mtcnt,
                 mesh_verts=np.concatenate(mv) if mv else np.zeros(0), mesh_tris=np.concatenate(mt) if mt else np.zeros(0))
    with open(path, "w") as f:
        for k, v in items.items():
            v = np.asarray(v).ravel()
            f.write("%s %d\n%s\n" % (k, len(v), " ".join(repr(float(x)) for x in v)))


def read_host_unit_dump(path):
    """-> meta dict, list of costs {kind, t, value, contacts: [dict]}, list of sweep hits."""
    meta, costs, sweep = None, [], []
    for line in open(path):
        w = line.split()
        if w[0] == "meta":
            meta = dict(T=int(w[1]), K=int(w[2]), calls_per_timestep=int(w[3]), calls_prefetched=int(w[4]), prefetch_hits=int(w[5]))
        elif w[0] == "cost":
            costs.append(dict(kind=w[1], t=int(w[2]), value=float(w[3]), n=int(w[4]), contacts=[]))
        elif w[0] == "contact":
            v = [float(x) for x in w[7:20]]
            nco = int(w[21])
            pairs = w[22:22 + 2 * nco]
            costs[-1]["contacts"].append(dict(
                idA=int(w[1][1:]), childA=int(w[2]), idB=int(w[3][1:]), childB=int(w[4]), cctype=int(w[5]), time=np.float32(w[6]),
                dist=v[0], n=v[1:4], ptA=v[4:7], ptB=v[7:10], ptB1=v[10:13], const=float(w[20]),
                vars=[int(x) for x in pairs[0::2]], coeffs=[float(x) for x in pairs[1::2]]))
        elif w[0] == "sweep":
            sweep.append(dict(idA=int(w[1][1:]), childA=int(w[2]), idB=int(w[3][1:]), childB=int(w[4]), toi=np.float32(w[5]), n=[float(x) for x in w[6:9]]))
    return meta, costs, sweep


def add_ref_scene(scene, name):
    from trajopt_amd import synth
    return synth.add_ref_scene(scene, name)


def benchmark_trajectories(rb, rng, n, T=11):
    from trajopt_amd import synth
    return synth.benchmark_trajectories(rb, rng, n, T)
