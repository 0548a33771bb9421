"""GPU suite, part 2: every way the engine can be configured computes the SAME slab, bit for bit, and the rarely taken
host / fall-back paths agree with the oracle.

  * engine variants (environment switches read by tohc_create): pre-test off, one stream, enqueues cut into unit
    ranges on four lanes, host-side canonical sort, 16-lane-group GJK, one EPA group per wavefront, two passes sharing one penetration tail,
    EPA runs repeated on the full pools;
  * several appended enqueues in one slab (the lanes' ordering events);
  * a robot with two joints on one optimisation column (mimic joint): k_finish's lane-per-column fall-back;
  * a robot with more than 64 optimisation columns: the same fall-back and tohc_fetch_hinges' wide column kernel.
"""
import os

import numpy as np
import pytest

from oracle import bindings as B
from trajopt_amd import model as M, synth
from tests import util

pytestmark = pytest.mark.gpu

VARIANTS = [{}, {"TOHC_PRETEST": "0"}, {"TOHC_STREAMS": "1"}, {"TOHC_CHUNKS": "2", "TOHC_STREAMS": "4"}, {"TOHC_HOST_SORT": "1"},
            {"TOHC_GJK_GROUP": "16"}, {"TOHC_EPA_BLOCK": "16"}, {"TOHC_GROUP": "64"}, {"TOHC_MERGE_TAIL": "1"},
            {"TOHC_EPA_SMALL": "5,10"},   # EPA's first-attempt pools so small that most runs are repeated on the full pools
            {"TOHC_DEFER_EXACT": "0"},    # the exact stage of the pre-test inside k_broadphase instead of in k_exact
            {"TOHC_GJK_SPLIT": "0"},      # every GJK run ends where it started (k_gjk) ...
            {"TOHC_GJK_SPLIT": "1"},      # ... or nearly every one goes on in k_gjk_cont (default: after 6 passes)
            {"TOHC_PEN_CAP": "700"},      # shards of the penetration hand-over queue so small that the largest class of pairs overflows
                                          # its four and falls back to the blocks' shards (k_tail_pen files pairs by GJK2 class)
            {"TOHC_GJK_ADAPT": "0"}, {"TOHC_HEADS_FIRST": "0"}]


def _world():
    """Full-body robot next to a 400-hull soup and a few boxes: separated, touching and deeply penetrating pairs."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="fullbody")
    synth.convex_soup(s, n_hulls=400, seed=5, extent=1.2, center=(1.1, 0.0))
    ids, boxT = synth.add_boxes(s, [((.85, .55, .01), synth.T7((1.1, 0, .635))), ((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))], name="box")
    s.finalize(); rb.finalize()
    static_T = synth.static_poses(s, ids, boxT)
    return s, rb, static_T


def _run(env, s, rb, static_T, dofs, Bp, Tt, allowed=None):
    from trajopt_amd import engine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = engine.Engine(0)       # the switches are read here
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        e.upload_scene(s, static_T)
        e.upload_robot(rb)
        if allowed is None:   # IgnoreZeroStateSelfCollisions (collision_checker.cpp:41-56): robot pairs touching at rest
            e.set_contact_distance(0.1)
            rec0, _ = e.links_vs_all_batch(np.zeros((1, rb.K)), 1, 1, filter_mask=M.GROUP_ROBOT, want_jacobians=False)
            allowed = np.ones((s.n_objects, s.n_objects), np.uint8)
            allowed[rec0["idA"], rec0["idB"]] = 0
            allowed[rec0["idB"], rec0["idA"]] = 0
        e.set_allowed_matrix(allowed)
        e.set_contact_distance(0.09)
        e.set_capacity(1 << 22, 1 << 20)
        e.stage_dofs(dofs, Bp, Tt)
        e.enqueue_links_vs_all(-1, True, append=False)
        e.enqueue_cast_vs_all(1, True, append=True)
        e.enqueue_links_vs_all(M.GROUP_KINBODY, True, append=True)   # a third and a fourth pass in the same slab
        e.enqueue_cast_vs_all(2, True, append=True)
        rec, g = e.fetch()
        c = e.counters()
        return rec, g, c, allowed
    finally:
        e.close()


def test_engine_variants_compute_the_same_slab(built):
    s, rb, static_T = _world()
    Bp, Tt = 6, 100            # 600 units per pass: enough for the unit-range split to engage
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(50 + b), Tt) for b in range(Bp)])
    ref_rec, ref_g, ref_c, allowed = _run({}, s, rb, static_T, dofs, Bp, Tt)
    assert len(ref_rec) > 1000 and ref_c["epa_evals"] > 0 and ref_c["gjk_evals"] < ref_c["pair_evals"]
    for env in VARIANTS[1:]:
        rec, g, c, _ = _run(env, s, rb, static_T, dofs, Bp, Tt, allowed)
        tag = str(env)
        assert len(rec) == len(ref_rec), tag
        # the canonical order leaves only exact duplicates (a robot-robot pair seen from both query links in the same
        # GJK roles) free to swap, and those are bitwise equal
        assert np.array_equal(rec, ref_rec), tag
        assert np.array_equal(g, ref_g), tag
        assert c["pair_evals"] == ref_c["pair_evals"] and c["contacts"] == ref_c["contacts"], tag
        if env.get("TOHC_PRETEST") == "0":
            assert c["gjk_evals"] == c["pair_evals"]
        if "TOHC_DEFER_EXACT" in env:   # the same test in either place: the same pairs reach k_gjk
            assert c["gjk_evals"] == ref_c["gjk_evals"], tag
        if "TOHC_PEN_CAP" in env:
            assert c["pen_refiled"] > 0 and ref_c["pen_refiled"] == 0, (c["pen_refiled"], ref_c["pen_refiled"])
        if "TOHC_EPA_SMALL" in env:
            assert c["epa_redone"] > 0.05 * c["epa_evals"] > 0, (c["epa_redone"], c["epa_evals"])
    # and the slab is right: one problem against the oracle, pass by pass
    w = B.oracle_world(s, cap=1 << 17)
    w.set_contact_distance(0.09)
    w.set_allowed(allowed)
    b = 2
    sel = ref_rec["problem"] == b
    rec_b, g_b = ref_rec[sel], ref_g[sel]
    o1, g1 = w.linearize(rb, static_T, dofs[b])
    o2, g2 = w.linearize(rb, static_T, dofs[b], cast=True)
    o3, g3 = w.linearize(rb, static_T, dofs[b], mask=M.GROUP_KINBODY)
    o4, g4 = w.linearize(rb, static_T, dofs[b], cast=True, gap=2)
    disc, cast = rec_b["cctype"] == 0, rec_b["cctype"] != 0
    assert disc.sum() == len(o1) + len(o3) and cast.sum() == len(o2) + len(o4)
    # same (t, pair) keys with multiplicity, same distances
    def keys(t, a, ca, bb, cb, dist):
        return sorted(zip(t.tolist(), a.tolist(), ca.tolist(), bb.tolist(), cb.tolist(), dist.tolist()))
    for m, os_ in ((disc, (o1, o3)), (cast, (o2, o4))):
        oc = np.concatenate([o["c"] for o in os_]); ot = np.concatenate([o["t"] for o in os_])
        r = rec_b[m]
        mine = keys(r["t"], r["idA"], r["childA"], r["idB"], r["childB"], r["dist"])
        assert mine == keys(ot, oc["objA"], oc["childA"], oc["objB"], oc["childB"], oc["dist"].astype(np.float64))
    w.close()


def test_unit_broadphase_equals_per_query_broadphase(built):
    """A small world (arm over a table and a box): with enough units (default: 1024; here forced with TOHC_UNIT_BP=1) ONE
    wavefront per unit does all its box tests (k_broadphase_unit); TOHC_UNIT_BP=0 sends them through k_broadphase's
    wavefront per (unit, query child).  Same slab, same pair-evaluation count."""
    s = M.Scene()
    rb = synth.pr2_like(s, seed=0, active="rightarm")
    ids, boxT = synth.add_boxes(s, [synth.TABLE_BOX, ((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))])
    s.finalize(); rb.finalize()
    static_T = synth.static_poses(s, ids, boxT)
    Bp, Tt = 16, 10
    rng = np.random.default_rng(21)
    dofs = np.stack([synth.straight_line(rb.q_robot[1:8] + rng.uniform(-0.5, 0.5, 7), rb.q_robot[1:8] + rng.uniform(-0.9, 0.9, 7), Tt) for _ in range(Bp)])
    ref_rec, ref_g, ref_c, allowed = _run({"TOHC_UNIT_BP": "1"}, s, rb, static_T, dofs, Bp, Tt)
    rec, g, c, _ = _run({"TOHC_UNIT_BP": "0"}, s, rb, static_T, dofs, Bp, Tt, allowed)
    assert len(rec) == len(ref_rec) > 100
    assert np.array_equal(rec, ref_rec) and np.array_equal(g, ref_g)
    assert c["pair_evals"] == ref_c["pair_evals"] and c["gjk_evals"] == ref_c["gjk_evals"]


def test_mimic_joint_falls_back_to_column_walk(built):
    """Two revolute joints of one chain driven by the same robot DOF (mult 1 and -0.5): their Jacobian columns add
    before the product with the normal, which only the lane-per-column epilogue does (DevRobot.fast_jac = 0)."""
    from trajopt_amd import engine
    s = M.Scene()
    r = M.RobotModel()
    base = r.add_link(-1, name="base")
    dims = [(0.3, 0.04, 0.04, 0.3), (0.25, 0.03, 0.03, 0.25), (0.2, 0.03, 0.03, 0.2)]
    parent, off = base, 0.0
    for i, (hx, hy, hz, cx) in enumerate(dims):
        obj = s.add_object([(s.add_box(hx, hy, hz), synth.T7((cx, 0, 0)))], M.GROUP_ROBOT, name="arm%d" % i)
        dof, mult = (0, 1.0) if i == 0 else ((1, 1.0) if i == 1 else (0, -0.5))   # link 2 mimics joint 0
        parent = r.add_link(parent, synth.T7((off, 0, 0.1 * i)), M.J_REVOLUTE, (0, 0, 1), dof=dof, mult=mult, offs=0.1 * i, obj=obj, name="arm%d" % i)
        off = 2 * hx
    r.set_active([0, 1])
    ids, boxT = synth.add_boxes(s, [((.3, .3, .3), synth.T7((1.0, 0.5, 0.1))), ((.05, .6, .2), synth.T7((0.5, -0.3, 0.1)))], name="obst")
    s.finalize(); r.finalize()
    static_T = synth.static_poses(s, ids, boxT)
    traj = synth.straight_line([-0.6, -0.8], [0.9, 0.7], 12)
    e = engine.Engine(0)
    try:
        e.upload_scene(s, static_T)
        e.upload_robot(r)
        e.set_contact_distance(0.15)
        w = B.oracle_world(s)
        w.set_contact_distance(0.15)
        n = 0
        for cast in (False, True):
            rec, g = e.cast_vs_all_batch(traj, 1, len(traj), 1) if cast else e.links_vs_all_batch(traj, 1, len(traj))
            orec, og = w.linearize(r, static_T, traj, cast=cast)
            util.assert_same_linearization(rec, g, orec, og, "mimic cast=%s" % cast)
            n += len(rec)
        assert n > 0
        w.close()
    finally:
        e.close()


def test_wide_robot_more_than_64_columns(built):
    """K = 70 active joints (> TOHC_FAST_JAC_MAXK = TOHC_HINGE_MAXK = 64): k_finish runs its lane-per-column epilogue
    (linearizeAt; no LDS rows are requested for it) and tohc_fetch_hinges its one-thread-per-variable column kernel
    (k_hinge_cols_wide).  A planar-ish snake of 70 revolute joints with alternating axes, one box per link, over a
    table and between two posts: engine vs oracle, hinge block vs the host assembly (bit for bit)."""
    from trajopt_amd import collision_checker as CC, engine
    s = M.Scene()
    r = M.RobotModel()
    parent = r.add_link(-1, name="base")
    NJ = 70
    for i in range(NJ):
        obj = s.add_object([(s.add_box(0.02, 0.012, 0.012), synth.T7((0.02, 0, 0)))], M.GROUP_ROBOT, name="seg%d" % i)
        parent = r.add_link(parent, synth.T7((0.04 if i else 0.0, 0, 0.5 if i == 0 else 0.0)), M.J_REVOLUTE, (0, 0, 1) if i % 2 == 0 else (0, 1, 0),
                            dof=i, obj=obj, name="seg%d" % i)
    r.set_active(range(NJ))
    ids, boxT = synth.add_boxes(s, [((1.5, 1.0, 0.01), synth.T7((1.2, 0, 0.44))), ((.05, .05, .3), synth.T7((1.0, 0.12, 0.6))), ((.05, .05, .3), synth.T7((1.9, -0.12, 0.6)))], name="obst")
    s.finalize(); r.finalize()
    assert r.K == NJ > 64
    static_T = synth.static_poses(s, ids, boxT)
    rng = np.random.default_rng(8)
    Bp, Tt = 2, 6
    dofs = np.stack([synth.straight_line(rng.uniform(-0.03, 0.03, NJ), rng.uniform(-0.05, 0.05, NJ), Tt) for _ in range(Bp)])
    e = engine.Engine(0)
    try:
        e.upload_scene(s, static_T)
        e.upload_robot(r)
        # neighbouring segments touch by construction: exclude what touches at the zero pose, like IgnoreZeroStateSelfCollisions
        e.set_contact_distance(0.1)
        rec0, _ = e.links_vs_all_batch(np.zeros((1, NJ)), 1, 1, filter_mask=M.GROUP_ROBOT, want_jacobians=False)
        allowed = np.ones((s.n_objects, s.n_objects), np.uint8)
        allowed[rec0["idA"], rec0["idB"]] = 0
        allowed[rec0["idB"], rec0["idA"]] = 0
        e.set_allowed_matrix(allowed)
        d = 0.08
        e.set_contact_distance(d)
        w = B.oracle_world(s)
        w.set_contact_distance(d)
        w.set_allowed(allowed)
        n = 0
        for cast in (False, True):
            rec, g = e.cast_vs_all_batch(dofs, Bp, Tt, 1) if cast else e.links_vs_all_batch(dofs, Bp, Tt)
            assert g.shape == (len(rec), 2, NJ)
            for b in range(Bp):
                orec, og = w.linearize(r, static_T, dofs[b], cast=cast)
                sel = rec["problem"] == b
                util.assert_same_linearization(rec[sel], g[sel], orec, og, "wide K cast=%s b=%d" % (cast, b))
                n += len(orec)
        assert n > 100
        w.close()
        # the hinge block of the wide robot (k_hinge_cols_wide) against the host assembly
        e.stage_dofs(dofs, Bp, Tt)
        e.enqueue_linearization(-1, 1, True)
        rec, g = e.fetch()
        h = e.fetch_hinges(0.05, 0.03)
        ref = CC.bpmpd_collision_block(rec, g, Bp, Tt, NJ, 1, 0.05, 0.03)
        for k in ("B", "T", "K", "gap", "n_rows", "nnz"):
            assert h[k] == ref[k], k
        for k in ("row_off", "rhs", "dist", "row_t", "col_off", "col_idx", "col_val"):
            assert h[k].tobytes() == ref[k].tobytes(), k
        assert h["nnz"] > 0
    finally:
        e.close()


def test_fetch_device_and_device_all_gather(built):
    """tohc_fetch_device hands out the slab in canonical order as it lies in device memory -- the same bytes tohc_fetch
    copies to the host -- and multi_gpu.all_gather_slabs_device feeds it to RCCL without a trip through host memory
    (here a one-rank `nccl` group on the one GPU; the two-rank exchange itself is tested over gloo on the CPU)."""
    import socket
    import torch
    import torch.distributed as dist
    from trajopt_amd import engine, multi_gpu
    s, rb, static_T = _world()
    Bp, Tt = 2, 20
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(70 + b), Tt) for b in range(Bp)])
    e = engine.Engine(0)
    try:
        e.upload_scene(s, static_T)
        e.upload_robot(rb)
        e.set_contact_distance(0.09)
        e.stage_dofs(dofs, Bp, Tt)
        e.enqueue_links_vs_all(-1, True, append=False)
        e.enqueue_cast_vs_all(1, True, append=True)
        d_rec, d_g, n, rowd = e.fetch_device()
        rec, g = e.fetch()
        assert n == len(rec) > 100 and rowd == 2 * rb.K and d_rec and d_g
        dev = torch.device("cuda", 0)
        got = torch.as_tensor(multi_gpu._DeviceBytes(d_rec, n * 152), device=dev).cpu().numpy()
        assert got.tobytes() == rec.tobytes()
        gotg = torch.as_tensor(multi_gpu._DeviceBytes(d_g, n * rowd * 8), device=dev).cpu().numpy()
        assert gotg.tobytes() == np.ascontiguousarray(g).tobytes()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        try:
            counts, recs, grads = multi_gpu.all_gather_slabs_device(e, dist)
            assert counts.tolist() == [n] and recs.is_cuda and grads.is_cuda
            assert recs[0, :n].cpu().numpy().tobytes() == rec.tobytes()
            assert np.array_equal(grads[0, :n].cpu().numpy().reshape(n, 2, rb.K), g)
        finally:
            dist.destroy_process_group()
        # without gradient rows the device slab has none either
        e.enqueue_links_vs_all(-1, False, append=False)
        _, d_g0, n0, rowd0 = e.fetch_device()
        assert n0 > 0 and d_g0 == 0 and rowd0 == 0
    finally:
        e.close()


def test_enqueue_linearization_equals_the_two_enqueues(built):
    """tohc_enqueue_linearization = tohc_enqueue_links_vs_all(append 0) + tohc_enqueue_cast_vs_all(append 1): the same slab
    and the same counters, with the two passes' broadphases in one launch (k_broadphase_both) or, where the engine's
    set-up does not allow that (TOHC_FUSE_BP=0, one stream, an enqueue cut into unit ranges), one after the other."""
    from trajopt_amd import engine
    s, rb, static_T = _world()
    Bp, Tt = 5, 30
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(80 + b), Tt) for b in range(Bp)])
    keys = ("aabb_tests", "pair_evals", "gjk_evals", "epa_evals", "contacts")
    for env in ({}, {"TOHC_FUSE_BP": "0"}, {"TOHC_STREAMS": "1"}, {"TOHC_CHUNKS": "2", "TOHC_STREAMS": "4"}, {"TOHC_MERGE_TAIL": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            e = engine.Engine(0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        try:
            e.upload_scene(s, static_T)
            e.upload_robot(rb)
            e.set_contact_distance(0.09)
            e.set_capacity(1 << 22, 1 << 20)
            e.stage_dofs(dofs, Bp, Tt)
            for mask, gap in ((-1, 1), (M.GROUP_KINBODY, 3), (-1, Tt)):   # (gap = T: no cast unit at all)
                e.enqueue_links_vs_all(mask, True, append=False)
                e.enqueue_cast_vs_all(gap, True, append=True)
                r0, g0 = e.fetch()
                c0 = e.counters()
                e.enqueue_linearization(mask, gap, True)
                r1, g1 = e.fetch()
                c1 = e.counters()
                tag = "%s mask %d gap %d" % (env, mask, gap)
                assert len(r0) > (1000 if gap < Tt else 100), tag
                assert np.array_equal(r0, r1) and np.array_equal(g0, g1), tag
                assert all(c0[k] == c1[k] for k in keys), (tag, [(k, c0[k], c1[k]) for k in keys])
            # a third pass appended to a linearization, like to any slab
            e.enqueue_linearization(-1, 1, True)
            e.enqueue_links_vs_all(M.GROUP_KINBODY, True, append=True)
            r2, _ = e.fetch()
            e.enqueue_links_vs_all(-1, True, append=False)
            e.enqueue_cast_vs_all(1, True, append=True)
            e.enqueue_links_vs_all(M.GROUP_KINBODY, True, append=True)
            r3, _ = e.fetch()
            assert np.array_equal(r2, r3), env
        finally:
            e.close()


def test_kernel_timing_off_same_slab_and_span(built):
    """tohc_set_kernel_timing(0) only leaves the per-stage event records out: same slab, same work counters, the span of
    the call is still measured and the per-stage sums stand still."""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0, n_hulls=512, seed=3)
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(70 + g), 9) for g in range(2)])
    cc = CC.HIPCollisionChecker(s, static_T, None)
    cc.SetRobot(rb); cc.SetContactDistance(d)
    eng = cc.eng
    eng.stage_dofs(dofs, 2, 9)
    eng.enqueue_linearization(-1, 1, True)
    rec, g = eng.fetch()
    c_on = eng.counters()
    eng.set_kernel_timing(False)
    eng.enqueue_linearization(-1, 1, True)
    rec2, g2 = eng.fetch()
    c_off = eng.counters()
    assert len(rec) > 0 and rec.tobytes() == rec2.tobytes() and g.tobytes() == g2.tobytes()
    assert all(c_on[k] == c_off[k] for k in ("pair_evals", "gjk_evals", "epa_evals", "contacts", "aabb_tests"))
    assert c_off["ms_last_call"] > 0 and c_on["cum_ms_gjk"] > 0
    assert all(c_off[k] == c_on[k] for k in ("cum_ms_broadphase", "cum_ms_gjk", "cum_ms_epa", "cum_ms_finish"))
    eng.set_kernel_timing(True)
    eng.enqueue_linearization(-1, 1, True)
    eng.sync()
    assert eng.counters()["cum_ms_gjk"] > c_on["cum_ms_gjk"]
    eng.close()


def test_small_batches_switch_gjk_mapping_same_slab(built):
    """A pass that held few pairs last time runs k_gjk<16> (16 lanes per pair) instead of k_gjk<1> the next time the same
    pass is enqueued (tohc_engine::gjk_adapt; TOHC_GJK_ADAPT=0 keeps one lane per pair): the first linearization of an
    engine and every later one give the same slab as an engine that never switches, and the same work counters."""
    from trajopt_amd import collision_checker as CC
    s, rb, static_T, d = synth.config_fullbody_soup(npts=0, n_hulls=512, seed=3)
    dofs = np.stack([synth.fullbody_trajectory(rb, np.random.default_rng(80 + g), 12) for g in range(2)])
    out = {}
    for adapt in ("1", "0"):
        old = os.environ.get("TOHC_GJK_ADAPT")
        os.environ["TOHC_GJK_ADAPT"] = adapt
        try:
            cc = CC.HIPCollisionChecker(s, static_T, None)
        finally:
            if old is None:
                os.environ.pop("TOHC_GJK_ADAPT", None)
            else:
                os.environ["TOHC_GJK_ADAPT"] = old
        cc.SetRobot(rb); cc.SetContactDistance(d)
        eng = cc.eng
        eng.stage_dofs(dofs, 2, 12)
        runs = []
        for _ in range(3):
            eng.enqueue_linearization(-1, 1, True)
            rec, g = eng.fetch()
            c = eng.counters()
            runs.append((rec.tobytes(), g.tobytes(), tuple(c[k] for k in ("pair_evals", "gjk_evals", "epa_evals", "contacts"))))
        assert len(rec) > 100 and 0 < c["gjk_evals"] <= 16384   # (few enough pairs for the switch)
        out[adapt] = runs
        eng.close()
    for r in out["1"] + out["0"]:
        assert r == out["0"][0]
