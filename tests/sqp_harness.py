"""TEST INFRASTRUCTURE, not product code: a small trust-region SQP around the collision path.

sco (BasicTrustRegionSQP) and its BPMPD back end are out of scope -- BPMPD ships in the reference only as
a prebuilt binary, which is never run here.  To check the north-star's last correctness gate ("the final SQP
trajectories match") and the reference's cast-cost-unit test (src/trajopt/test/cast-cost-unit.cpp:37-81: optimise
data/box_cast_test.json, expect 0 continuous collisions) the loop of src/sco/optimizers.cpp:220-400 is restated here
over numpy, with the convex subproblem solved by a dense ADMM (OSQP-style) QP solver.  The same harness is driven once by
the HIP engine and once by the CPU oracle; only the source of the collision linearization differs.

Problem shape (problem_description.cpp): variables = the T x K trajectory; start fixed by bounds (basic_info.start_fixed),
"joint" constraint = the last row pinned (JointConstraintInfo::hatch adds exact linear equalities to the model: here
lb = ub); joint_vel cost = sum coeffs_k (x[t+1,k] - x[t,k])^2; collision cost = coeff * hinge(dist_pen - dist), one
hinge per contact, discrete and / or cast (collision_terms.cpp:196-223).
"""
import numpy as np
import scipy.linalg as sla


def solve_qp(P, q, A, l, u, iters=20000, eps=1e-10):
    """min 1/2 z'Pz + q'z  s.t.  l <= Az <= u   (ADMM as in OSQP, adaptive rho, dense)."""
    n, m = len(q), len(l)
    sigma, alpha = 1e-6, 1.6
    eq = (u - l) < 1e-12
    rho = 0.1
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)

    def factor(rho):
        rv = np.where(eq, 1e3 * rho, rho)
        return rv, sla.cho_factor(P + sigma * np.eye(n) + A.T @ (rv[:, None] * A))

    rv, F = factor(rho)
    for k in range(iters):
        rhs = sigma * x - q + A.T @ (rv * z - y)
        xt = sla.cho_solve(F, rhs)
        zt = A @ xt
        x = alpha * xt + (1 - alpha) * x
        zh = alpha * zt + (1 - alpha) * z
        zn = np.clip(zh + y / rv, l, u)
        y = y + rv * (zh - zn)
        z = zn
        if k % 25 == 24:
            Ax = A @ x
            rp = np.abs(Ax - z).max(initial=0)
            rd = np.abs(P @ x + q + A.T @ y).max(initial=0)
            if rp < eps and rd < eps:
                break
            if k % 100 == 99:
                sp = rp / max(np.abs(Ax).max(initial=0), np.abs(z).max(initial=0), 1e-12)
                sd = rd / max(np.abs(P @ x).max(initial=0), np.abs(A.T @ y).max(initial=0), np.abs(q).max(initial=0), 1e-12)
                new = rho * np.sqrt(sp / max(sd, 1e-16))
                if new > 5 * rho or new < rho / 5:
                    rho = float(np.clip(new, 1e-6, 1e6))
                    rv, F = factor(rho)
    return x


class TrajProblem:
    """linearize(traj, cast) -> (t[n], dist[n], time[n] float32, const0[n], const1[n], grads[n,2,K]); see the adapters below."""

    def __init__(self, traj0, vel_coeffs, dist_pen, coll_coeff, linearize, discrete=True, cast=True, end_fixed=True):
        self.T, self.K = traj0.shape
        self.traj0 = traj0.copy()
        self.vel = np.broadcast_to(np.asarray(vel_coeffs, float), (self.K,)).copy()
        self.dist_pen, self.coeff, self.lin = dist_pen, coll_coeff, linearize
        self.kinds = [c for c, on in ((False, discrete), (True, cast)) if on]
        self.lb = np.full((self.T, self.K), -np.inf)
        self.ub = np.full((self.T, self.K), np.inf)
        self.lb[0] = self.ub[0] = traj0[0]
        if end_fixed:
            self.lb[-1] = self.ub[-1] = traj0[-1]
        # joint_vel as a quadratic form (JointVelCost, trajectory_costs.cpp)
        n = self.T * self.K
        D = np.zeros(((self.T - 1) * self.K, n))
        for t in range(self.T - 1):
            for k in range(self.K):
                D[t * self.K + k, t * self.K + k], D[t * self.K + k, (t + 1) * self.K + k] = -1, 1
        w = np.tile(self.vel, self.T - 1)
        self.Pvel = 2 * D.T @ (w[:, None] * D)
        self.n_lin = 0

    def vel_cost(self, x):
        return float(0.5 * x @ self.Pvel @ x)

    def collision(self, x, jac):
        """-> value, hinge rows [(a (sparse over the full x), b)] with hinge(b + a.x)"""
        traj = x.reshape(self.T, self.K)
        val, rows = 0.0, []
        for cast in self.kinds:
            t, dist, time, c0, c1, g = self.lin(traj, cast)
            self.n_lin += 1
            val += float(np.maximum(0.0, self.dist_pen - dist).sum() * self.coeff)
            if not jac:
                continue
            for i in range(len(t)):
                a = np.zeros(self.T * self.K)
                ti = int(t[i])
                if cast:
                    w0, w1 = float(np.float32(1) - time[i]), float(time[i])       # collision_terms.cpp:73-74
                    a[ti * self.K:(ti + 1) * self.K] = -w0 * g[i, 0]
                    a[(ti + 1) * self.K:(ti + 2) * self.K] = -w1 * g[i, 1]
                    b = self.dist_pen - (w0 * c0[i] + w1 * c1[i])
                else:
                    a[ti * self.K:(ti + 1) * self.K] = -g[i, 0]
                    b = self.dist_pen - c0[i]
                rows.append((a, b))
        return val, rows


def optimize(prob, max_iter=40, improve_ratio_threshold=.2, min_approx_improve_frac=.001, trust_box_size=.1, min_trust_box_size=1e-4,
             min_approx_improve=1e-4, trust_shrink_ratio=.1, trust_expand_ratio=1.5, log=None):
    """BasicTrustRegionSQP::optimize (optimizers.cpp:220-400) with OptimizeProblem's settings (problem_description.cpp:244-257).
    There are no nonlinear constraints in these problems, so the merit loop runs once."""
    n = prob.T * prob.K
    x = np.clip(prob.traj0.ravel(), prob.lb.ravel(), prob.ub.ravel())           # getClosestFeasiblePoint
    coll, _ = prob.collision(x, False)
    cost = prob.vel_cost(x) + coll
    status = "iteration limit"
    for it in range(1, max_iter + 1):
        _, rows = prob.collision(x, True)                                        # convexifyCosts
        m = len(rows)
        P = np.zeros((n + m, n + m)); P[:n, :n] = prob.Pvel
        q = np.concatenate([np.zeros(n), np.full(m, prob.coeff)])
        A = np.zeros((n + 2 * m, n + m)); A[:n, :n] = np.eye(n)
        lo, hi = np.zeros(n + 2 * m), np.zeros(n + 2 * m)
        for i, (a, b) in enumerate(rows):                                        # hinge: s >= 0, s >= b + a.x
            A[n + i, n + i] = 1; lo[n + i], hi[n + i] = 0, np.inf
            A[n + m + i, :n] = a; A[n + m + i, n + i] = -1; lo[n + m + i], hi[n + m + i] = -np.inf, -b
        converged = False
        while trust_box_size >= min_trust_box_size:
            lo[:n] = np.maximum(prob.lb.ravel(), x - trust_box_size)             # setTrustBoxConstraints
            hi[:n] = np.minimum(prob.ub.ravel(), x + trust_box_size)
            z = solve_qp(P, q, A, lo, hi)
            new_x = np.clip(z[:n], lo[:n], hi[:n])
            model = prob.vel_cost(new_x) + prob.coeff * sum(max(0.0, b + a @ new_x) for a, b in rows)
            new_coll, _ = prob.collision(new_x, False)                           # evaluateCosts
            new_cost = prob.vel_cost(new_x) + new_coll
            approx, exact = cost - model, cost - new_cost
            if log is not None:
                log.append((it, trust_box_size, cost, approx, exact))
            if approx < min_approx_improve or approx / cost < min_approx_improve_frac:
                converged = True
                break
            if exact < 0 or exact / approx < improve_ratio_threshold:
                trust_box_size *= trust_shrink_ratio
            else:
                x, cost = new_x, new_cost
                trust_box_size *= trust_expand_ratio
                break
        if converged or trust_box_size < min_trust_box_size:
            status = "converged"
            break
    return x.reshape(prob.T, prob.K), cost, status


# ---- the two sources of the linearization ---------------------------------------------------------------------------
def hip_linearizer(cc):
    """trajopt_amd.collision_checker.HIPCollisionChecker -> linearize callable (the GPU path)."""
    def lin(traj, cast):
        if cast:
            rec, g = cc.eng.cast_vs_all_batch(traj, 1, len(traj), 1, True)
        else:
            rec, g = cc.eng.links_vs_all_batch(traj, 1, len(traj), -1, True)
        if g is None:
            g = np.zeros((0, 2, traj.shape[1]))
        return rec["t"], rec["dist"], rec["time"], rec["const0"], rec["const1"], g
    return lin


def oracle_linearizer(w, rb, static_T):
    """oracle.bindings world -> linearize callable (the CPU checker), records in the engine's canonical order."""
    def lin(traj, cast):
        orec, og = w.linearize(rb, static_T, traj, cast=cast)
        c = orec["c"]
        o = np.lexsort((c["childB"], c["objB"], c["childA"], c["objA"], orec["t"]))
        orec, og, c = orec[o], og[o], c[o]
        return orec["t"], c["dist"].astype(float), c["time"], orec["const0"], orec["const1"], og
    return lin
