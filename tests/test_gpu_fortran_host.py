"""-m gpu: the C ABI driven from Fortran.  roms_trunk_mgh_amd/fortran/roms_hip_demo.F90 (a small host program of
this repository, not ROMS) registers one tile's arrays through c_loc, issues main3d's calls through the
ISO_C_BINDING module roms_hip_mod and writes the prognostic fields back; the result must equal the run driven
from Python (ctypes) bit for bit -- same library, same calls, two bindings."""
import ctypes as C
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from roms_trunk_mgh_amd import abi, ana, hip, main3d

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "roms_trunk_mgh_amd", "fortran")


@pytest.mark.skipif(shutil.which("flang") is None, reason="flang not installed")
@pytest.mark.parametrize("config,river", [("UPWELLING", False), ("BENCHMARK_TINY", False), ("UPWELLING", True)])
def test_fortran_host_equals_python_host(tmp_path, config, river):
    """river: a basin with point sources (LuvSrc and LwSrc) -- roms_hip_set_sources through the Fortran interface"""
    nsteps = 4
    libdir = os.path.join(ROOT, "roms_trunk_mgh_amd")
    r = subprocess.run(["flang", "-c", os.path.join(FDIR, "roms_hip_mod.F90"), "-o", "m.o"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["flang", os.path.join(FDIR, "roms_hip_demo.F90"), "m.o", "-L" + libdir, "-lroms_hip",
                        "-Wl,-rpath," + libdir, "-o", "demo"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    st = ana.make_tile(config, perturb=1.0, overrides={"EWperiodic": False} if river else None)
    if river:
        import util
        src = util.river_sources(st, "all")
    # the state as the Fortran host receives it: bounds and parameter blocks as they lie in memory, then every
    # registered field in the order of include/roms_fields.def (Fortran order, as the module arrays are)
    with open(tmp_path / "state.bin", "wb") as f:
        for blk in (st.b, st.p):
            raw = bytes(memoryview(blk))
            f.write(struct.pack("q", len(raw)))
            f.write(raw)
        f.write(struct.pack("q", len(abi.FIELDS)))
        for name, _kind, _grp in abi.FIELDS:
            a = np.asfortranarray(st[name])
            f.write(struct.pack("qq", abi.FIELD_ID[name], a.size))
            f.write(a.tobytes(order="F"))
        if river:                 # SOURCES(ng): Isrc, Jsrc, Dsrc, Qbar, Qsrc(Nsrc,N), Tsrc(Nsrc,N,NT), LtracerSrc
            q = src.qsrc()
            f.write(struct.pack("qqq", src.n, q.size, st.b.NT))
            for a in (src.Isrc, src.Jsrc, src.Dsrc, src.Qbar):
                f.write(a.tobytes())
            f.write(q.tobytes(order="F"))
            f.write(src.Tsrc.tobytes(order="F"))
            f.write(src.LtracerSrc.tobytes())
    r = subprocess.run([str(tmp_path / "demo"), "state.bin", "result.bin", str(nsteps)], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "roms_hip_demo:" in r.stdout
    # the same run from Python
    be = hip.RomsHip(st)
    try:
        m = main3d.Main3D(be, physics=False, diagnostics=True)
        m.initial()
        m.run(nsteps)
        be.to_host(["zeta", "ubar", "vbar", "u", "v", "t"])
    finally:
        be.close()
    with open(tmp_path / "result.bin", "rb") as f:
        nout, indx1, nnew = struct.unpack("qqq", f.read(24))
        assert (indx1, nnew) == (m.indx1, m.s.nnew)
        for _ in range(nout):
            fid, cnt = struct.unpack("qq", f.read(16))
            got = np.frombuffer(f.read(8 * cnt), dtype=np.float64)
            name = abi.FIELDS[fid][0]
            want = np.asfortranarray(st[name]).reshape(-1, order="F")
            assert np.array_equal(got, want), name
        d12 = np.frombuffer(f.read(96), dtype=np.float64)
    assert np.array_equal(d12, m.last_diag)
    assert float(np.abs(st["u"]).max()) > 1e-6


MPI_HOME = "/opt/conda"


@pytest.mark.skipif(shutil.which("flang") is None or not os.path.exists(os.path.join(MPI_HOME, "bin", "mpiexec")),
                    reason="flang or MPICH not installed")
@pytest.mark.parametrize("ntI,ntJ,config", [(2, 1, "BENCHMARK_TINY"), (2, 2, "SEAMOUNT")])
def test_fortran_mpi_host_tiles_equal_single_tile(tmp_path, ntI, ntJ, config):
    """One MPI rank per tile (all on the one GPU of the box), the halo exchange of the library carried by the
    host's MPI_Isend / MPI_Irecv / MPI_Waitall through roms_hip_set_halo_relay -- where the reference's
    mp_exchange posts them.  Every tile (owned and ghost points) must equal the single-tile Python-driven run."""
    nsteps, world = 3, ntI * ntJ
    libdir = os.path.join(ROOT, "roms_trunk_mgh_amd")
    r = subprocess.run(["flang", "-c", os.path.join(FDIR, "roms_hip_mod.F90"), "-o", "m.o"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["flang", "-I" + os.path.join(MPI_HOME, "include"), os.path.join(FDIR, "roms_hip_demo_mpi.F90"), "m.o",
                        "-L" + libdir, "-lroms_hip", "-L" + os.path.join(MPI_HOME, "lib"), "-lmpifort", "-lmpi",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(MPI_HOME, "lib"), "-o", "demo_mpi"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # single-tile run from Python first (so that at most world + 1 processes hold the GPU)
    ref = ana.make_tile(config, perturb=1.0)
    be = hip.RomsHip(ref)
    try:
        m = main3d.Main3D(be)
        m.initial()
        m.run(nsteps)
        be.to_host()
    finally:
        be.close()
    tiles = []
    for rank in range(world):
        st = ana.make_tile(config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=1.0)
        tiles.append(st)
        with open(tmp_path / f"state_{rank}.bin", "wb") as f:
            for blk in (st.b, st.p):
                raw = bytes(memoryview(blk))
                f.write(struct.pack("q", len(raw)))
                f.write(raw)
            f.write(struct.pack("q", len(abi.FIELDS)))
            for name, _kind, _grp in abi.FIELDS:
                a = np.asfortranarray(st[name])
                f.write(struct.pack("qq", abi.FIELD_ID[name], a.size))
                f.write(a.tobytes(order="F"))
    r = subprocess.run([os.path.join(MPI_HOME, "bin", "mpiexec"), "-n", str(world), str(tmp_path / "demo_mpi"), "state_", "result_",
                        str(nsteps), str(ntI), str(ntJ)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rb = ref.b
    for rank, st in enumerate(tiles):
        b = st.b
        with open(tmp_path / f"result_{rank}.bin", "rb") as f:
            nout, indx1, nnew = struct.unpack("qqq", f.read(24))
            assert (indx1, nnew) == (m.indx1, m.s.nnew)
            for _ in range(nout):
                fid, cnt = struct.unpack("qq", f.read(16))
                name = abi.FIELDS[fid][0]
                a = np.frombuffer(f.read(8 * cnt), dtype=np.float64).reshape(st[name].shape, order="F")
                ni, nj = a.shape[0], a.shape[1]
                i0, j0 = b.LBi - rb.LBi, b.LBj - rb.LBj
                want = ref[name][i0:i0 + ni, j0:j0 + nj]
                own = (slice(b.Istr - b.LBi, b.Iend - b.LBi + 1), slice(b.Jstr - b.LBj, b.Jend - b.LBj + 1))
                assert np.array_equal(a[own], want[own]), (name, rank)
                if name in ("zeta", "t", "Hz", "W"):      # rho-type: every ghost point is defined
                    iv = min(ni, rb.Lm + rb.NghostPoints - b.LBi + 1)
                    jv = min(nj, rb.Mm + 1 - b.LBj + 1)
                    assert np.array_equal(a[:iv, :jv], want[:iv, :jv]), (name, rank, "ghost points differ")
