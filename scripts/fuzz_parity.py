"""Parity fuzz: the HIP engine against the CPU oracle on randomly drawn scenes, for a time budget instead of a fixed seed
list -- what tests/test_gpu_parity.py does for a dozen seeds, here for as many as fit (GPU box: the oracle's CPU time is
what bounds it, so several worker processes each drive their own engine on the one GPU and their own oracle on a core).

  python scripts/fuzz_parity.py --seconds 600 --workers 12 --first-seed 100000 --out gpurun_out/fuzz.json

Kinds of case (seed % 6):
  0  random poses-mode scene of hulls + boxes, 4 pose sets: LinksVsAll, CastVsAll, AllVsAll at d = 0 and 0.09   -- bit for bit
  1  the same with z-cylinders                                                                                -- bit for bit
  2  the same with ~45 % spheres                                                                              -- bit for bit
  3  robot mode: full-body robot against a random 300-hull soup (42-vertex hulls every other case) + two boxes, 2 problems
     x 6 timesteps, discrete + cast linearization: contact set and values bit for bit, gradient rows to 1e-5 relative
  4  ContinuousCheckTrajectory on explicit poses (k_sweep), 5 pose sets, d = 0 and 0.05: hit set, fraction, normal, point
     bit for bit (oracle in portable-trig mode, as in tests/test_gpu_parity.py)
  5  RayCastCollision, 2000 rays: fractions bit for bit
Kinds 0 - 3 also hold the engine's pair-evaluation count (the unit of the headline metric) equal to the oracle's.
The oracle is the checker, never the thing measured.  A mismatch is recorded with its seed and the run goes on.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_case(eng, seed):
    import numpy as np
    from oracle import bindings as B
    from trajopt_amd import model as M, synth
    from tests import util
    rng = np.random.default_rng(seed)
    kind = seed % 6
    stats = {"contacts": 0, "pair_evals": 0, "gjk_evals": 0, "epa_evals": 0}

    def book():
        c = eng.counters()
        for k in ("pair_evals", "gjk_evals", "epa_evals"):
            stats[k] += int(c[k])

    if kind < 3:
        spread = float(rng.choice([0.25, 0.3, 0.35, 0.45, 0.6, 1.0]))
        if kind == 2:
            s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=6, round_prob=0.9, spheres=True)
        else:
            s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=5, round_prob=0.5 if kind == 1 else 0.0)
        nsets = 4
        sets = np.stack([objT] + [np.array([synth.random_T7(rng, -spread, spread) for _ in range(len(objT))]) for _ in range(nsets - 1)])
        T1 = sets[:, robot].copy()
        T1[..., 4:] += rng.normal(size=T1[..., 4:].shape) * 0.15
        T1[..., :4] += rng.normal(size=T1[..., :4].shape) * 0.2
        T1[..., :4] /= np.linalg.norm(T1[..., :4], axis=-1, keepdims=True)
        eng.upload_scene(s, objT)
        w = B.oracle_world(s)
        try:
            for d in (0.0, 0.09):
                eng.set_contact_distance(d)
                w.set_contact_distance(d)
                rec = eng.links_vs_all_poses(sets, robot); book()
                pe_d = int(eng.counters()["pair_evals"])
                recc = eng.cast_vs_all_poses(sets, robot, T1); book()
                pe_c = int(eng.counters()["pair_evals"])
                c0 = w.counter(0)
                for k in range(nsets):
                    util.assert_same_contacts(util.records_to_contacts(rec[rec["problem"] == k]), w.links_vs_all(sets[k], robot), "disc set %d d=%g" % (k, d))
                c1 = w.counter(0)
                for k in range(nsets):
                    util.assert_same_contacts(util.records_to_contacts(recc[recc["problem"] == k]),
                                              w.cast_vs_all(sets[k], robot, sets[k][robot], T1[k]), "cast set %d d=%g" % (k, d))
                # the unit of the headline metric: both sides count the same pair evaluations (box-cull survivors)
                assert (pe_d, pe_c) == (c1 - c0, w.counter(0) - c1), "pair evaluations differ d=%g: engine %s oracle %s" % (d, (pe_d, pe_c), (c1 - c0, w.counter(0) - c1))
                ava = eng.all_vs_all_poses(sets[0]); book()
                util.assert_same_contacts(util.records_to_contacts(ava), w.all_vs_all(sets[0]), "all-vs-all d=%g" % d)
                stats["contacts"] += len(rec) + len(recc) + len(ava)
        finally:
            w.close()
    elif kind >= 4:
        spread = float(rng.choice([0.4, 0.6, 0.8, 1.5]))
        sub = (seed // 6) % 3   # hulls + boxes / + cylinders / + spheres
        s, objT, robot = synth.random_scene(rng, spread=spread, n_static=8, n_robot=5, round_prob=(0.0, 0.3, 0.8)[sub], spheres=sub == 2)
        eng.upload_scene(s, objT)
        w = B.oracle_world(s)
        try:
            if kind == 4:
                nst = 5
                traj = np.stack([np.array([synth.random_T7(rng, -spread, spread) for _ in robot]) for _ in range(nst)])
                sets = np.tile(objT[None], (nst, 1, 1))
                sets[:, robot] = traj
                w.set_trig_mode(1)
                try:
                    for d in (0.0, 0.05):
                        eng.set_contact_distance(d)
                        w.set_contact_distance(d)
                        rec = eng.continuous_check_poses(sets, robot)
                        hits = w.continuous_check(objT, robot, traj)
                        rk = sorted((int(r["t"]), int(r["idA"]), int(r["childA"]), int(r["idB"]), int(r["childB"]), float(np.float32(r["time"])),
                                     tuple(r["n"].tolist()), tuple(r["ptA"].tolist())) for r in rec)
                        hk = sorted((int(h["seg"]), int(h["obj"]), int(h["child"]), int(h["hitObj"]), int(h["hitChild"]), float(h["fraction"]),
                                     tuple(np.asarray(h["n"], np.float64).tolist()), tuple(np.asarray(h["pt"], np.float64).tolist())) for h in hits)
                        assert rk == hk, "sweeps differ d=%g: %s" % (d, sorted(set(rk) ^ set(hk))[:4])
                        stats["contacts"] += len(rk)
                finally:
                    w.set_trig_mode(0)
            else:
                n = 2000
                a = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
                b = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
                a[:50] = objT[rng.integers(0, len(objT), 50), 4:7]
                b[50:60] = a[50:60]
                fr, fo = eng.ray_cast_poses(objT, a, b), w.ray_cast(objT, a, b)
                assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32)), "ray fractions differ at %s" % np.nonzero(fr != fo)[0][:8]
                stats["contacts"] += int((fo < 1).sum())
        finally:
            w.close()
    else:
        s = M.Scene()
        rb = synth.pr2_like(s, seed=int(rng.integers(0, 1000)), active="fullbody", npts=0 if (seed // 6) % 2 else 120)
        synth.convex_soup(s, n_hulls=300, seed=int(rng.integers(0, 1 << 30)), extent=float(rng.choice([0.9, 1.2, 1.6])), center=(1.1, 0.0),
                          npts=0 if (seed // 6) % 2 else 120)
        ids, boxT = synth.add_boxes(s, [((.85, .55, .01), synth.T7((1.1, 0, .635))), ((.2, .3, .4), synth.T7((0.6, -0.6, 0.4), (0, 0, 1), 0.5))], name="box")
        s.finalize(); rb.finalize()
        static_T = synth.static_poses(s, ids, boxT)
        Bp, Tt = 2, 6
        dofs = np.stack([synth.fullbody_trajectory(rb, rng, Tt) for _ in range(Bp)])
        eng.upload_scene(s, static_T)
        eng.upload_robot(rb)
        d = float(rng.choice([0.02, 0.065, 0.09]))
        eng.set_contact_distance(d)
        w = B.oracle_world(s, cap=1 << 17)
        try:
            w.set_contact_distance(d)
            eng.stage_dofs(dofs, Bp, Tt)
            eng.enqueue_linearization(-1, 1, True)
            rec, g = eng.fetch()
            book()
            pe = int(eng.counters()["pair_evals"])
            c0 = w.counter(0)
            for b in range(Bp):
                sel = rec["problem"] == b
                disc = sel & (rec["cctype"] == 0)
                cast = sel & (rec["cctype"] != 0)
                orec, og = w.linearize(rb, static_T, dofs[b])
                util.assert_same_linearization(rec[disc], g[disc], orec, og, "robot disc b=%d d=%g" % (b, d))
                orec, og = w.linearize(rb, static_T, dofs[b], cast=True)
                util.assert_same_linearization(rec[cast], g[cast], orec, og, "robot cast b=%d d=%g" % (b, d))
            assert pe == w.counter(0) - c0, "pair evaluations differ: engine %d oracle %d" % (pe, w.counter(0) - c0)
            stats["contacts"] += len(rec)
        finally:
            w.close()
    return stats


def worker(wid, nworkers, first_seed, seconds, q):
    try:
        from trajopt_amd import engine
        eng = engine.Engine(0)
        eng.set_capacity(1 << 21, 1 << 18)
        t_end = time.time() + seconds
        seed = first_seed + wid
        tot = {"cases": 0, "contacts": 0, "pair_evals": 0, "gjk_evals": 0, "epa_evals": 0, "by_kind": [0, 0, 0, 0, 0, 0]}
        bad = []
        t_last = time.time()
        while time.time() < t_end:
            try:
                st = one_case(eng, seed)
                for k, v in st.items():
                    tot[k] += v
                tot["by_kind"][seed % 6] += 1
            except AssertionError as ex:
                bad.append({"seed": seed, "what": str(ex)[:400]})
            tot["cases"] += 1
            seed += nworkers
            if time.time() - t_last > 45:
                q.put(("progress", wid, dict(tot), len(bad)))
                t_last = time.time()
        eng.close()
        q.put(("done", wid, tot, bad))
    except Exception:
        q.put(("crash", wid, traceback.format_exc(), []))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--first-seed", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz.json"))
    a = ap.parse_args()
    ctx = mp.get_context("spawn")   # (children start fresh: no GPU state is inherited)
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(i, a.workers, a.first_seed, a.seconds, q)) for i in range(a.workers)]
    t0 = time.time()
    for p in procs:
        p.start()
    done, tot, bad, crashes = 0, {}, [], []
    while done < a.workers:
        try:
            msg = q.get(timeout=a.seconds + 600)
        except Exception:
            crashes.append("timed out waiting for the workers")
            break
        if msg[0] == "progress":
            print("[%.0f s] worker %d: %d cases, %d contacts compared, %d mismatching cases" % (time.time() - t0, msg[1], msg[2]["cases"], msg[2]["contacts"], msg[3]), flush=True)
            continue
        done += 1
        if msg[0] == "crash":
            crashes.append(msg[2])
            continue
        for k, v in msg[2].items():
            tot[k] = [x + y for x, y in zip(tot.get(k, [0, 0, 0, 0, 0, 0]), v)] if isinstance(v, list) else tot.get(k, 0) + v
        bad += msg[3]
    for p in procs:
        p.join(timeout=30)
    out = {"seconds": time.time() - t0, "workers": a.workers, "first_seed": a.first_seed, "totals": tot, "mismatching_cases": bad, "crashes": crashes,
           "what": "engine vs oracle: contact sets and values bit for bit (poses mode: LinksVsAll, CastVsAll, AllVsAll; robot mode: + gradient rows to 1e-5 relative)"}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("seconds", "totals")}), "mismatches:", len(bad), "crashes:", len(crashes), flush=True)
    for b in bad[:10]:
        print("  MISMATCH", b)
    for c in crashes[:3]:
        print("  CRASH", c)
    sys.exit(1 if (bad or crashes) else 0)


if __name__ == "__main__":
    main()
