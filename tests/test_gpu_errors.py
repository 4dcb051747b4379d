"""-m gpu: the error convention of the C ABI: whatever the library cannot do it
refuses with a non-zero return code and a message in roms_hip_last_error() -- it never aborts the host
program, never falls back to another path, and stays usable afterwards."""
import numpy as np
import pytest

import util
from roms_trunk_mgh_amd import abi, ana, hip

pytestmark = pytest.mark.gpu


def test_unsupported_boundary_condition_is_refused():
    st = ana.make_tile("UPWELLING", perturb=1.0)
    st.p.lbc_south = 99                                    # no such code (enum roms_lbc ends at LBC_REDUCED = 10)
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("pre_step3d", util.step_idx())
        assert "not implemented" in str(e.value)
        with pytest.raises(RuntimeError):
            h.call("step3d_t", util.step_idx())
    finally:
        h.close()
    # a valid code on a variable it is not defined for: Shchepetkin (ubar / vbar only) as the side's default, which
    # also applies to zeta, u, v and t
    st = ana.make_tile("UPWELLING", perturb=1.0)
    st.p.lbc_south = abi.LBC["Shc"]
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("pre_step3d", util.step_idx())
        assert "not implemented" in str(e.value)
    finally:
        h.close()
    # a condition that exists, for a variable it is not defined for: Flather on the free surface
    st = ana.make_tile("UPWELLING", perturb=1.0)
    st.p.lbc[abi.LBS["north"]][abi.LBV["zeta"]] = abi.LBC["Fla"]
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError):
            h.call("step2d", util.step_idx())
    finally:
        h.close()


def test_unsupported_advection_pair_is_refused_and_library_stays_usable():
    # MPDATA horizontally with C4 vertically: the reference allows it, this library restates the pair only
    st = ana.make_tile("UPWELLING", perturb=1.0, overrides={"Hadv": "U3", "Vadv": "C4"})
    st.p.Hadv[0] = abi.ADV["MPDATA"]
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("step3d_t", util.step_idx())
        assert "not implemented" in str(e.value)
    finally:
        h.close()
    # a fresh context afterwards works and gives the usual answer
    import oracle
    st0 = util.prepared_state("UPWELLING")
    st_o, st_h = st0.copy(), st0.copy()
    oracle.Oracle(st_o).call("omega", util.step_idx())
    h = hip.RomsHip(st_h)
    try:
        h.call("omega", util.step_idx())
        h.to_host()
    finally:
        h.close()
    assert np.array_equal(st_h["W"], st_o["W"])


def test_snapshot_of_unknown_field_and_double_begin_are_refused():
    st = ana.make_tile("UPWELLING", perturb=1.0)
    h = hip.RomsHip(st)
    try:
        h.snapshot_begin(["zeta"])
        with pytest.raises(RuntimeError) as e:
            h.snapshot_begin(["zeta"])
        assert "in flight" in str(e.value)
        h.snapshot_end()
        h.snapshot_begin(["zeta"])
        h.snapshot_end()
    finally:
        h.close()


def test_fields_of_absent_options_may_stay_unregistered():
    """An application without MASKING / TS_DIF4 / UV_VIS4 has no masks and no biharmonic coefficients: the library
    keeps all-water masks and zero coefficients and the step is the same, bit for bit.
    With the option set the field is required."""
    import numpy as np
    from roms_trunk_mgh_amd import main3d
    absent = ("rmask", "umask", "vmask", "pmask", "visc4_p", "visc4_r", "diff4")
    out = []
    for leave in ((), absent):
        st = ana.make_tile("UPWELLING", perturb=1.0)
        h = hip.RomsHip(st, leave_unregistered=leave)
        try:
            m = main3d.Main3D(h)
            m.initial()
            m.run(3)
            h.to_host()
        finally:
            h.close()
        out.append(st)
    for name in ("zeta", "u", "v", "t"):
        assert np.array_equal(out[0][name], out[1][name]), name
    st = ana.make_tile("UPWELLING", perturb=1.0, mask="island")
    h = hip.RomsHip(st, leave_unregistered=("umask",))
    try:
        with pytest.raises(RuntimeError) as e:
            h.call("set_depth", util.step_idx())
        assert "field not registered: umask" in str(e.value)
    finally:
        h.close()


def test_library_defaults_do_not_survive_a_parameter_change():
    """the all-water masks the library keeps for an application without MASKING must not be used after
    set_params switches MASKING on -- the entry has to refuse with 'field not registered'."""
    import ctypes as C
    st = util.prepared_state("UPWELLING")
    h = hip.RomsHip(st, leave_unregistered=("rmask", "umask", "vmask", "pmask"))
    try:
        h.call("omega", util.step_idx())                   # creates the defaults under masking = 0
        p2 = type(st.p).from_buffer_copy(st.p)
        p2.masking = 1
        assert h.l.roms_hip_set_params(C.byref(p2)) == 0
        with pytest.raises(RuntimeError) as e:
            h.call("omega", util.step_idx())
        assert "field not registered" in str(e.value)
        p2.masking = 0                                     # and back: usable again
        assert h.l.roms_hip_set_params(C.byref(p2)) == 0
        h.call("omega", util.step_idx())
    finally:
        h.close()


def test_point_sources_without_their_table_are_refused():
    """An application with rivers must not run without them: with LuvSrc or LwSrc set and no roms_hip_set_sources call
    every entry fails with a message; a table with a Dsrc outside 0, 1, 2 is refused, and so is a table handed to an
    application that has set neither switch."""
    import util
    from roms_trunk_mgh_amd import hip, sources
    st = util.prepared_state("UPWELLING")
    st.p = type(st.p).from_buffer_copy(st.p)
    for flag in (1, 2, 3):
        st.p.point_sources = flag
        h = hip.RomsHip(st)
        try:
            for entry in ("step2d", "step3d_t", "omega"):
                with pytest.raises(RuntimeError) as e:
                    h.call(entry, util.step_idx())
                assert "point sources" in str(e.value)
        finally:
            h.close()
    st.p.point_sources = 1
    N, NT = st.b.N, st.b.NT
    bad = sources.Sources([5], [5], [3.0], [10.0], np.full((1, N), 1.0 / N), np.zeros((1, N, NT)), np.ones(NT, dtype=np.int32))
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.set_sources(bad)
        assert "Dsrc" in str(e.value)
    finally:
        h.close()
    st.p.point_sources = 0
    ok = sources.Sources([5], [5], [0.0], [10.0], np.full((1, N), 1.0 / N), np.zeros((1, N, NT)), np.ones(NT, dtype=np.int32))
    h = hip.RomsHip(st)
    try:
        with pytest.raises(RuntimeError) as e:
            h.set_sources(ok)
        assert "point_sources" in str(e.value)
    finally:
        h.close()
