"""WET_DRY known-answer and property tests.  wetdry.F cannot be built from the reference here (it reaches mod_sources
-> mod_netcdf), and step2d / step3d_uv / pre_step3d are parity-unpinned anyway, so their WET_DRY blocks are checked
against facts that do not come from the C restatement:

* the mask rules of wetdry_mask_tile / wetdry_avg_mask_tile restated on whole numpy arrays from the rule in words
  (rho: wet = sea and total depth above Dcrit; u / v: 2 between two wet cells, 0 between two dry ones, +1 / -1 when
  only the lower- / higher-index cell is wet; psi: 1 with at least three wet neighbours, 2 with two on the same side),
  compared with what one step2d call leaves;
* the fast-step accumulation: a cell is wet for the baroclinic step only if every one of the 2*nfast barotropic calls
  found it wet; a one-sided face is open only to flow out of its wet cell (sign of DU_avg1 / DV_avg1);
* consequences on a drying beach: no transport through a face between two dry cells, no flow INTO the wet cell
  through a one-sided face, the total depth never below Dcrit by more than one fast step's transport, the volume
  of water conserved to round-off over 100 steps (closed walls, periodic channel), tracers bounded.
Each test runs on the oracle (CPU) and, with -m gpu, on the HIP path."""
import numpy as np
import pytest

from roms_trunk_mgh_amd import ana, main3d

import util

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]
BEACH = {"wet_dry": 1, "beach": 1, "zeta_amp": 0.3}


def _backend(kind, st):
    if kind == "hip":
        from roms_trunk_mgh_amd import hip
        return hip.RomsHip(st)
    import oracle
    return oracle.Oracle(st)


def _sync(be):
    if hasattr(be, "to_host"):
        be.to_host()


def _close(be):
    if hasattr(be, "close"):
        be.close()


def _flag(st, zeta):
    wd = np.where(zeta + st["h"] <= st.p.Dcrit + 1.0e-10, 0.0, 1.0) * st["rmask"]
    return wd


def _fast_masks(wd):
    """u / v / psi masks of the fast steps from the rho-point flag, from the rule in words."""
    m, s = slice(0, -1), slice(1, None)
    um = np.full_like(wd, np.nan)
    vm = np.full_like(wd, np.nan)
    pm = np.full_like(wd, np.nan)
    lo, hi = wd[m, :], wd[s, :]
    um[s, :] = np.where((lo == 1) & (hi == 1), 2.0, np.where((lo == 0) & (hi == 0), 0.0, np.where(lo == 1, 1.0, -1.0)))
    lo, hi = wd[:, m], wd[:, s]
    vm[:, s] = np.where((lo == 1) & (hi == 1), 2.0, np.where((lo == 0) & (hi == 0), 0.0, np.where(lo == 1, 1.0, -1.0)))
    a, b, c, d = wd[m, s] > 0.5, wd[s, s] > 0.5, wd[m, m] > 0.5, wd[s, m] > 0.5
    n = a.astype(int) + b.astype(int) + c.astype(int) + d.astype(int)
    diag = (a & d & ~b & ~c) | (b & c & ~a & ~d)
    pm[s, s] = np.where(n >= 3, 1.0, np.where((n == 2) & ~diag, 2.0, 0.0))
    return um, vm, pm


def _owned(st, a, gtype):
    """The part of a 2-D mask array the routine computes on one periodic-channel tile (wetdry.F:588-683)."""
    b = st.b
    i0 = (b.Istr if gtype in "up" else b.IstrR) - b.LBi
    j0 = (b.Jstr if gtype in "vp" else b.JstrR) - b.LBj
    return a[i0:b.IendR - b.LBi + 1, j0:b.JendR - b.LBj + 1]


@pytest.mark.parametrize("kind", BACKENDS)
def test_fast_step_masks_follow_the_rule(kind):
    st = ana.make_tile("UPWELLING", perturb=1.0, overrides=BEACH)
    be = _backend(kind, st)
    try:
        m = main3d.Main3D(be)
        m.initial()
        m.step()                                  # a consistent state with all time levels set
        _sync(be)
        s = util.step_idx(iic=2, iif=3, pred=1, kstp=2, krhs=1, knew=3)
        s.ntfirst = 1
        zk = st["zeta"][:, :, s.kstp - 1].copy()
        avg0 = st["rmask_wet_avg"].copy()
        if hasattr(be, "to_device"):
            be.to_device()
        be.call("step2d", s)
        _sync(be)
    finally:
        _close(be)
    wd = _flag(st, zk)
    assert 0 < wd.sum() < wd.size and (wd * st["rmask"] == wd).all()
    um, vm, pm = _fast_masks(wd)
    assert np.array_equal(_owned(st, st["rmask_wet"], "r"), _owned(st, wd, "r"))
    assert np.array_equal(_owned(st, st["umask_wet"], "u"), _owned(st, um, "u"))
    assert np.array_equal(_owned(st, st["vmask_wet"], "v"), _owned(st, vm, "v"))
    assert np.array_equal(_owned(st, st["pmask_wet"], "p"), _owned(st, pm, "p"))
    assert {-1.0, 0.0, 1.0, 2.0} <= set(np.unique(_owned(st, st["umask_wet"], "u"))) | set(np.unique(_owned(st, st["vmask_wet"], "v")))
    # the running sum of the rho-point flag (not the first predictor step: it accumulates)
    assert np.array_equal(_owned(st, st["rmask_wet_avg"], "r"), _owned(st, avg0 + wd, "r"))


@pytest.mark.parametrize("kind", BACKENDS)
def test_baroclinic_masks_are_the_strictest_of_the_fast_steps(kind):
    """After LOOP_2D: rmask_wet = 1 only where all 2*nfast calls found the cell wet (recorded here by stepping the
    loop by hand), one-sided faces open only for DU_avg1 / DV_avg1 out of the wet cell, lone ponds closed."""
    st = ana.make_tile("UPWELLING", perturb=1.0, overrides=BEACH)
    be = _backend(kind, st)
    try:
        m = main3d.Main3D(be)
        m.initial()
        for _ in range(3):
            m.step()
        _sync(be)
    finally:
        _close(be)
    nf2 = 2.0 * st.p.nfast
    avg = st["rmask_wet_avg"]
    wd = np.trunc(avg / nf2)
    assert set(np.unique(_owned(st, avg, "r"))) <= set(np.arange(0.0, nf2 + 1.0))
    assert 0 < (_owned(st, avg, "r") % nf2 != 0).sum(), "no cell changed state during the fast steps: weak test"
    assert np.array_equal(_owned(st, st["rmask_wet"], "r"), _owned(st, wd, "r"))
    um, vm, pm = _fast_masks(wd)
    DU, DV = st["DU_avg1"], st["DV_avg1"]
    # face between two wet cells: 1; two dry cells: 0; one-sided: 1 only if the transport leaves the wet cell
    uw = np.where(um == 2, 1.0, np.where(um == 0, 0.0, np.where(um * np.where(np.signbit(DU), -1.0, 1.0) > 0, 1.0, 0.0)))
    uw = np.where((DU == 0.0) & (um != 2), 0.0, uw)
    vw = np.where(vm == 2, 1.0, np.where(vm == 0, 0.0, np.where(vm * np.where(np.signbit(DV), -1.0, 1.0) > 0, 1.0, 0.0)))
    vw = np.where((DV == 0.0) & (vm != 2), 0.0, vw)
    assert np.array_equal(_owned(st, st["umask_wet"], "u"), _owned(st, uw, "u"))
    assert np.array_equal(_owned(st, st["vmask_wet"], "v"), _owned(st, vw, "v"))
    assert np.array_equal(_owned(st, st["pmask_wet"], "p"), _owned(st, pm, "p"))
    for r, w, f in (("rmask", "rmask_wet", "rmask_full"), ("umask", "umask_wet", "umask_full"), ("vmask", "vmask_wet", "vmask_full")):
        g = r[0]
        assert np.array_equal(_owned(st, st[f], g), _owned(st, st[w] * st[r], g))
    assert (_owned(st, st["pmask_full"], "p") == 2.0).all()            # as written: MAX(pmask_wet*pmask, 2)


def _volume(st, lev):
    b = st.b
    own = (slice(b.Istr - b.LBi, b.Iend - b.LBi + 1), slice(b.Jstr - b.LBj, b.Jend - b.LBj + 1))
    return float((((st["zeta"][:, :, lev] + st["h"]) / (st["pm"] * st["pn"]))[own]).sum())


@pytest.mark.parametrize("kind", BACKENDS)
def test_drying_beach_100_steps(kind):
    st = ana.make_tile("UPWELLING", perturb=1.0, overrides=BEACH)
    be = _backend(kind, st)
    b, p = st.b, st.p
    try:
        m = main3d.Main3D(be)
        m.initial()
        m.step()
        _sync(be)
        # the first step floors the free surface at Dcrit - h (ini_fields.F:951-957); volume is conserved from here on
        v0 = _volume(st, 0)
        tmin, tmax = float(st["t"][..., 0].min()), float(st["t"][..., 0].max())
        changes, prev = 0, st["rmask_wet"].copy()
        if hasattr(be, "to_device"):
            be.to_device()
        for n in range(99):
            m.step()
            if n % 9 == 8 or n == 98:
                _sync(be)
                rw, uw, vw = st["rmask_wet"], st["umask_wet"], st["vmask_wet"]
                changes += int((rw != prev).sum())
                prev = rw.copy()
                lev = m.s.nnew - 1
                u, v = st["u"][:, :, :, lev], st["v"][:, :, :, lev]
                # no flow through closed faces
                assert (u[uw == 0.0] == 0.0).all() and (v[vw == 0.0] == 0.0).all()
                D = st["zeta"][:, :, 0] + st["h"]
                assert float(_owned(st, D, "r").min()) > 0.5 * p.Dcrit
                assert np.isfinite(st["t"]).all()
                if hasattr(be, "to_device"):
                    be.to_device()
        _sync(be)
    finally:
        _close(be)
    assert changes > 50, "the shoreline did not move: weak test"
    # zeta(:,:,1) and (:,:,2) hold the fast-time average after step3d_uv's coupling (set_zeta); the flux-form update
    # conserves the volume of every wet or dry cell
    v1 = _volume(st, 0)
    assert abs(v1 - v0) <= 1.0e-9 * abs(v0), (v0, v1)
    # tracers stay within (a little beyond, MPDATA is not used here) their initial range
    span = tmax - tmin
    assert tmin - 0.2 * span <= float(st["t"][..., 0].min()) and float(st["t"][..., 0].max()) <= tmax + 0.2 * span
