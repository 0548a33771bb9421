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
