"""tohc_ray_cast_poses_noalloc (include/trajopt_hipcc.h): the allocation-free variant of tohc_ray_cast_poses, called
through the C-ABI with caller-owned device memory (torch tensors), against the allocating entry point: closest hit
fractions BIT FOR BIT, on two calls in a row, the second with fewer rays and other poses (the library's fixed working
buffer keeps the first call's contents), plus the size limits.  Collected last (the file name sorts after the other
tests): a fault here must not hide them."""
import numpy as np
import pytest

from trajopt_amd import synth

pytestmark = pytest.mark.gpu


def test_noalloc_ray_cast_equals_allocating_entry_point(built):
    import torch
    from trajopt_amd import engine
    assert torch.cuda.is_available()
    e = engine.Engine(0)
    try:
        rng = np.random.default_rng(77)
        spread = 0.8
        s, objT, robot = synth.random_scene(rng, spread=spread, n_static=10, n_robot=6, round_prob=0.3)
        e.upload_scene(s, objT)
        e.set_contact_distance(0.05)
        hits = 0
        for n, poses in ((6000, objT), (1500, np.array([synth.random_T7(rng, -spread, spread) for _ in range(len(objT))]))):
            a = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
            b = rng.uniform(-1.5 * spread, 1.5 * spread, (n, 3))
            a[:20] = poses[rng.integers(0, len(poses), 20), 4:7]   # rays that start inside objects
            b[20:25] = a[20:25]                                    # zero-length rays
            dT = torch.tensor(np.ascontiguousarray(poses, np.float64), device="cuda:0")
            da = torch.tensor(a, dtype=torch.float64, device="cuda:0")
            db = torch.tensor(b, dtype=torch.float64, device="cuda:0")
            out = torch.full((n + 8,), -7.0, dtype=torch.float32, device="cuda:0")   # 8 guard words past the end
            torch.cuda.synchronize()
            # the allocation-free call FIRST: on the first round the engine is fresh -- no allocating ray cast has run on it,
            # its grow-only ray / pose buffers do not exist yet -- so the call stands on the library's fixed array alone
            e.ray_cast_poses_noalloc(dT.data_ptr(), da.data_ptr(), db.data_ptr(), n, out.data_ptr())
            got = out.cpu().numpy()
            want = e.ray_cast_poses(poses, a, b)
            assert np.array_equal(got[:n].view(np.uint32), want.view(np.uint32)), "n=%d: %d fractions differ" % (n, int((got[:n] != want).sum()))
            assert (got[n:] == -7.0).all()
            hits += int((want < 1).sum())
        assert hits > 200
        # the allocating entry point is unchanged by the calls above
        assert np.array_equal(e.ray_cast_poses(poses, a, b).view(np.uint32), want.view(np.uint32))
        # limits: rejected with a clear error, nothing runs
        with pytest.raises(engine.EngineError, match="rays in one call") as ex:
            e.ray_cast_poses_noalloc(dT.data_ptr(), da.data_ptr(), db.data_ptr(), 65537, out.data_ptr())
        assert ex.value.code == -3
        with pytest.raises(engine.EngineError):
            e.ray_cast_poses_noalloc(0, da.data_ptr(), db.data_ptr(), 4, out.data_ptr())
        e.ray_cast_poses_noalloc(dT.data_ptr(), 0, 0, 0, 0)   # no rays: nothing to do
    finally:
        e.close()
