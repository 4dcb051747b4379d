#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: simulated-days per wall-second of the ROMS
nonlinear 3-D time-stepping hot path on BENCHMARK3 (2048x256x30), on N MI355X.

A "step" is one full baroclinic step of the hot path:
set_massflux, rho_eos, omega, set_zeta, rhs3d (pre_step3d, prsgrd, t3dmix2,
rhs3d_tile, uv3dmix2), the 59-call barotropic step2d loop, set_depth,
step3d_uv, omega, step3d_t -- on synthetic (analytic) BENCHMARK inputs that are
resident in HBM before the timed region starts.  By default the per-step physics and
diagnostics the reference's BENCHMARK step carries (ana_srflux,
bulk_flux, set_vbc, lmd_vmix = KPP, wvelocity, diag with NINFO = 1) run on the device as well;
--no-physics holds their outputs fixed and times the section-8a hot path alone.

    python bench.py --gpus N --steps K --warmup W [--config BENCHMARK3 | BENCHMARK3_MPDATA | BENCHMARK1 ...]

For N > 1 the driver launches this file under torch.distributed.run, one rank
per GPU; the grid is split into NtileI x NtileJ = N tiles (2x1, 4x1, 4x2) and the
halo swaps run over RCCL inside libroms_hip.so (strong scaling: the global grid
is fixed).  Rank 0 prints ONE JSON line.

cpu_baseline (N = 1 only): the CPU port (oracle/) on ALL host cores the process may use, one
process per tile under mpiexec with the reference's two-phase MPI halo exchange
(oracle/mpi/oracle_mpi.c) -- the arrangement of the reference's MPI-Fortran build, which cannot be
linked here (no netCDF-Fortran).  The workers import neither torch nor HIP.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TILINGS = {1: (1, 1), 2: (2, 1), 4: (4, 1), 8: (4, 2)}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MPIEXEC = "/opt/conda/bin/mpiexec"
CPU_SHARE = 16                 # host cores that come with one GPU of the pool's boxes

# bench configurations: name -> (grid of ana.CONFIGS, NT, overrides)
MPDATA = {"Hadv": "MPDATA", "Vadv": "MPDATA"}
BENCH_CONFIGS = {
    # BASELINE.json configuration 5: BENCHMARK3 + 4 passive tracers, all six tracers advected with MPDATA
    "BENCHMARK3_MPDATA": ("BENCHMARK3", 6, MPDATA),
    "BENCHMARK1_MPDATA": ("BENCHMARK1", 6, MPDATA),
}


def make_tile(config, **kw):
    from roms_trunk_mgh_amd import ana
    if config in BENCH_CONFIGS:
        grid, NT, ov = BENCH_CONFIGS[config]
        return ana.make_tile(grid, NT=NT, overrides=ov, **kw)
    return ana.make_tile(config, **kw)


def pmc_traffic(config, gpus):
    """HBM-side bytes per launch of the roofline kernel, from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE).  Counters
    cannot be read from inside the bench; the figure holds for the configuration it was measured on
    (whole grid on one GPU) and is null otherwise.  Returns (bytes, source): `source` names the profile
    directory and the commit of the kernel source the counters were taken on, so that a figure that is
    older than the kernel shows."""
    if gpus != 1:
        return None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            k = json.load(f)[config]
        k = k.get("step3d_t") or k["k_step3d_t_pipe"]
        src = {"profile": "profiles/" + str(k.get("round")), "kernel_commit": k.get("commit"),
               "kernel_source_sha16": k.get("source_sha16"), "kernel_source_sha16_now": kernel_source_sha(config)}
        src["stale"] = bool(src["kernel_source_sha16"] and src["kernel_source_sha16"] != src["kernel_source_sha16_now"])
        return float(k["fetch_bytes"] + k["write_bytes"]), src
    except (OSError, KeyError, ValueError):
        return None, None


def kernel_source_sha(config):
    """sha256 (16 hex digits) of the source file of the roofline kernel of `config`."""
    import hashlib
    name = "k_mpdata.hip" if config.endswith("_MPDATA") else "k_step3d_t.hip"
    h = hashlib.sha256()
    for fn in (name, "k_mpdata_faces.inc") if name == "k_mpdata.hip" else (name,):
        try:
            with open(os.path.join(ROOT, "roms_trunk_mgh_amd", "csrc", fn), "rb") as f:
                h.update(f.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


# ----------------------------------------------------------------------------------------------
# CPU baseline: the oracle, one process per tile on all host cores, halos over MPI
# ----------------------------------------------------------------------------------------------
def cpu_tiling(nproc, Lm, Mm):
    """NtileI x NtileJ = nproc with the shortest tile perimeter (ties: more tiles along i, the long side)."""
    best = None
    for a in range(1, nproc + 1):
        if nproc % a:
            continue
        bq = nproc // a
        if a > Lm // 8 or bq > Mm // 8:
            continue
        cost = (Lm / a + Mm / bq, -a)
        if best is None or cost < best[0]:
            best = (cost, a, bq)
    return (best[1], best[2]) if best else (nproc, 1)


def _cpu_mpi_worker(config, nsteps, physics, tiling=None, outdir=None):
    """One MPI rank = one tile of the CPU oracle (started by mpiexec; no torch, no HIP).  `tiling` / `outdir`:
    used by tests/test_multitile_gloo.py, which checks this arrangement against the one-tile run."""
    import ctypes as C
    import numpy as np  # noqa: F401
    import oracle
    from roms_trunk_mgh_amd import abi, ana, main3d
    mpi = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_mpi.so"))
    mpi.oracle_mpi_wtime.restype = C.c_double
    mpi.oracle_mpi_max.restype = C.c_double
    mpi.oracle_mpi_max.argtypes = [C.c_double]
    rank = mpi.oracle_mpi_init()
    world = mpi.oracle_mpi_size()
    grid = BENCH_CONFIGS[config][0] if config in BENCH_CONFIGS else config
    ntI, ntJ = tiling or cpu_tiling(world, ana.CONFIGS[grid]["Lm"], ana.CONFIGS[grid]["Mm"])
    assert ntI * ntJ == world
    st = make_tile(config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=1.0)
    mpi.oracle_mpi_setup.argtypes = [C.POINTER(abi.Bounds)]
    mpi.oracle_mpi_setup(C.byref(st.b))
    lib = oracle.lib()
    HOOK = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_int)
    lib.oracle_set_exchange_hook.argtypes = [C.c_void_p]
    lib.oracle_set_exchange_hook(C.cast(mpi.oracle_mpi_exchange, C.c_void_p))
    be = oracle.Oracle(st)
    if physics and world > 1:
        tile_diag = be.diag
        dp = C.POINTER(C.c_double)
        mpi.oracle_mpi_allgather12.argtypes = [dp, dp]

        def global_diag(s_):
            v = tile_diag(s_)
            allv = np.zeros(12 * world)
            mpi.oracle_mpi_allgather12(v.ctypes.data_as(dp), allv.ctypes.data_as(dp))
            return main3d.reduce_diag(allv.reshape(world, 12))
        be.diag = global_diag
    m = main3d.Main3D(be, physics=physics, diagnostics=physics)
    m.initial()
    m.step()                      # first step (forward Euler branch) untimed
    mpi.oracle_mpi_barrier()
    t0 = mpi.oracle_mpi_wtime()
    m.run(nsteps)
    mpi.oracle_mpi_barrier()
    wall = mpi.oracle_mpi_max(mpi.oracle_mpi_wtime() - t0)
    ok = bool(np.isfinite(st["zeta"]).all())
    lib.oracle_set_exchange_hook(None)
    if outdir:
        bq = st.b
        np.savez(os.path.join(outdir, f"tile{rank}.npz"),
                 bounds=np.array([bq.Istr, bq.Iend, bq.Jstr, bq.Jend, bq.LBi, bq.LBj]),
                 **{k: st[k] for k in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz", "Akv", "tke")})
    if rank == 0:
        print("CPUBASE " + json.dumps({"wall": wall, "dt": st.p.dt, "tiling": f"{ntI}x{ntJ}", "ranks": world,
                                       "finite": ok}), flush=True)
    mpi.oracle_mpi_finalize()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_mpi(config, nsteps, physics, nproc, variant):
    """`nsteps` full steps of `config` on `nproc` host cores: mpiexec -n nproc, one oracle process per tile."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               ROMS_ORACLE_VARIANT=variant, PYTHONPATH=ROOT)
    cmd = [MPIEXEC, "-n", str(nproc), sys.executable, os.path.abspath(__file__), "--cpu-worker", config,
           str(nsteps), "1" if physics else "0"]
    sys.stderr.write(f"[bench] CPU baseline: mpiexec -n {nproc}, {variant} build, {nsteps} steps of {config} ...\n")
    sys.stderr.flush()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=360)
    for line in r.stdout.splitlines():
        if line.startswith("CPUBASE "):
            rec = json.loads(line[8:])
            rec["value"] = nsteps * rec["dt"] / 86400.0 / rec["wall"]
            return rec
    raise RuntimeError(f"mpiexec rc={r.returncode}: {r.stderr[-600:]}")


def cpu_baseline_single(config, nsteps, physics=True):
    """Fallback where mpiexec is missing: the oracle on ONE core, in this process."""
    import oracle
    from roms_trunk_mgh_amd import main3d
    st = make_tile(config, perturb=1.0)
    m = main3d.Main3D(oracle.Oracle(st), physics=physics, diagnostics=physics)
    m.initial()
    m.step()                      # first step (forward Euler branch) untimed
    t0 = time.perf_counter()
    m.run(nsteps)
    wall = time.perf_counter() - t0
    return nsteps * st.p.dt / 86400.0 / wall, wall


def cpu_baseline(config, physics, nsteps, ncores):
    """The cpu_baseline object of the bench line."""
    if os.path.exists(MPIEXEC) and os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle_mpi.so")):
        runs = {}
        for variant, flags in (("O2", "gcc -O2 -ffp-contract=off"), ("O3", "gcc -O3 -march=native")):
            try:
                runs[variant] = dict(cpu_baseline_mpi(config, nsteps, physics, ncores, variant), flags=flags)
            except Exception as e:            # the measured line must not depend on this leg
                sys.stderr.write(f"[bench] CPU baseline ({variant}) failed: {e!r}\n")
        if runs:
            best = max(runs.values(), key=lambda r: r["value"])
            return {"value": best["value"], "unit": "simulated-days/s", "cores": best["ranks"], "kind": "port",
                    "cpu_model": cpu_model(), "tiling": best["tiling"], "compiler_flags": best["flags"],
                    "by_flags": {r["flags"]: r["value"] for r in runs.values()},
                    "host_cores_visible": len(os.sched_getaffinity(0)),
                    "sample": f"{nsteps} full steps of {config} (after one untimed step) on {best['ranks']} host "
                              f"cores (the CPU share of a one-GPU box; the host shows "
                              f"{len(os.sched_getaffinity(0))}), one oracle/ process per tile ({best['tiling']}) "
                              f"under mpiexec, two-phase MPI halo exchange as mp_exchange.F "
                              f"({best['wall']:.1f} s wall); C restatement of the reference kernels"}
    v, w = cpu_baseline_single(config, min(nsteps, 3), physics)
    return {"value": v, "unit": "simulated-days/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "compiler_flags": "gcc -O2 -ffp-contract=off",
            "sample": f"{min(nsteps, 3)} full steps of {config} on ONE host core ({w:.1f} s): mpiexec or "
                      f"liboracle_mpi.so missing on this box"}


# ----------------------------------------------------------------------------------------------
# halo self-test (N > 1): one exchange of a field whose values encode their global index
# ----------------------------------------------------------------------------------------------
def halo_selftest(be, st, rank):
    """Fill wvel (a diagnostic output, rewritten every step) with f(i_global, j, k) on the points this tile
    owns and NaN elsewhere, run ONE roms_hip_exchange through the configured transport and check every
    ghost point that has a source tile.  Returns the number of wrong ghost values."""
    import ctypes as C
    import numpy as np
    from roms_trunk_mgh_amd import abi, hip
    b = st.b
    A = st["wvel"]
    nk = A.shape[2]
    ii = np.arange(b.LBi, b.UBi + 1)
    jj = np.arange(b.LBj, b.UBj + 1)
    gi = ((ii - 1) % b.Lm) + 1 if b.EWperiodic else ii
    f = (gi[:, None, None] + 4096.0 * jj[None, :, None] + 2.0 ** 24 * np.arange(nk)[None, None, :]).astype(np.float64)
    own_i = (ii >= b.Istr) & (ii <= b.Iend)
    own_j = (jj >= b.JstrR) & (jj <= b.JendR)
    own = own_i[:, None] & own_j[None, :]
    A[...] = np.where(own[:, :, None], f, np.nan)
    be.to_device(["wvel"])
    be._chk(be.l.roms_hip_exchange(abi.FIELD_ID["wvel"], 0), "exchange (halo self-test)")
    be.to_host(["wvel"])
    n = hip.tile_neighbors(rank, b.ntileI, b.ntileJ, b.NghostPoints, b.NghostPoints, b.EWperiodic, b.NSperiodic)
    ilo = b.Istr - (n["GrecvW"] if n["Wtile"] >= 0 else 0)
    ihi = b.Iend + (n["GrecvE"] if n["Etile"] >= 0 else 0)
    jlo = b.Jstr - (n["GrecvS"] if n["Stile"] >= 0 else 0)
    jhi = b.Jend + (n["GrecvN"] if n["Ntile"] >= 0 else 0)
    # rows without a tile beyond them (closed walls) carry the boundary rows JstrR / JendR of the W/E neighbours
    if n["Stile"] < 0:
        jlo = b.JstrR
    if n["Ntile"] < 0:
        jhi = b.JendR
    chk = ((ii >= ilo) & (ii <= ihi))[:, None] & ((jj >= jlo) & (jj <= jhi))[None, :]
    got, want = A[chk], f[chk]
    bad = int(np.count_nonzero(got != want))
    A[...] = 0.0
    be.to_device(["wvel"])
    return bad


def run_config5(args, device, steps=6, warmup=2):
    """BASELINE.json configuration 5 -- BENCHMARK3 + 4 passive tracers, all six advected with MPDATA -- on one
    GPU: the whole step with the reciprocal-sharing k_mp_adiff (roms_params_t.mpdata_fast = 1, as
    `--config BENCHMARK3_MPDATA` times it) and step3d_t alone with both variants (mpdata_fast = 0 is the one
    that is bit-identical to the oracle)."""
    import ctypes as C
    from roms_trunk_mgh_amd import hip, main3d
    config = "BENCHMARK3_MPDATA"
    st = make_tile(config, perturb=1.0)
    st.p.mpdata_fast = 1
    be = hip.RomsHip(st, rank=0, device=device)
    try:
        m = main3d.Main3D(be, physics=args.physics, diagnostics=args.physics)
        m.initial()
        for _ in range(warmup):
            m.step()
        be.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.step()
        be.sync()
        wall = time.perf_counter() - t0

        def step3d_t_ms(fast):
            p2 = type(st.p).from_buffer_copy(st.p)
            p2.mpdata_fast = fast
            be._chk(be.l.roms_hip_set_params(C.byref(p2)), "set_params")
            be.timing(False)
            m.step()
            be.timing(True)
            v = []
            for _ in range(4):
                m.step()
                v.append(be.last_ms("step3d_t"))
            be.timing(False)
            return sum(v) / len(v)
        t_fast = step3d_t_ms(1)
        t_exact = step3d_t_ms(0)
        import numpy as np
        be.to_host(["t"])
        ok = bool(np.isfinite(st["t"]).all())
    finally:
        be.close()
    b = st.b
    cells = (b.Iend - b.Istr + 1) * (b.Jend - b.Jstr + 1) * b.N
    comp = 8.0 * (4 * b.NT + 4) * cells
    multi = 8.0 * (12 * b.NT + 6) * cells
    traffic, src = pmc_traffic(config, 1)
    return {"workload": f"{config} {b.Lm}x{b.Mm}x{b.N} NT={b.NT}, all tracers MPDATA, "
                        + ("bulk fluxes + KPP + diagnostics every step" if args.physics else "fixed forcing"),
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall / steps,
            "value": steps * st.p.dt / 86400.0 / wall, "unit": "simulated-days/s", "mpdata_fast": 1,
            "step3d_t_ms": t_fast, "step3d_t_ms_exact": t_exact,
            "frac": comp / (t_fast * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_exact": comp / (t_exact * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "multipass_frac": multi / (t_fast * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": comp, "multipass_bytes": multi,
            "multipass_note": "8*(12*NT+6) B/cell, the round-2 definition (Ta, Ua, Va, Wa, beta_up, beta_dn materialised); "
                              "since round 3 beta_up / beta_dn stay in LDS",
            "traffic": traffic, "traffic_source": src, "finite": ok}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        a = sys.argv[2:]
        _cpu_mpi_worker(a[0], int(a[1]), a[2] == "1", tiling=(int(a[3]), int(a[4])) if len(a) > 4 else None,
                        outdir=a[5] if len(a) > 5 else None)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="BENCHMARK3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="host cores of the CPU baseline (default: min(cores this process may use, 16 = the CPU share of a one-GPU box))")
    ap.add_argument("--mpdata-exact", action="store_true",
                    help="MPDATA configurations: IEEE divisions in mpdata_adiff (bit-identical to the oracle) instead of "
                         "the refined reciprocals (within 1e-10 relative RMS of it, the default)")
    ap.add_argument("--loopback", action="store_true",
                    help="N = 1 only: hand the library an RCCL id so that the tile is its own western / eastern neighbour "
                         "and every periodic exchange of the step travels through pack -> ncclSend/ncclRecv -> unpack "
                         "(what a tile of an N-GPU run executes, measured on one GPU; not the headline configuration)")
    ap.add_argument("--graph-exchanges", choices=["auto", "on", "off"], default="auto",
                    help="several tiles over RCCL: replay LOOP_2D with its exchanges as one hipGraph (roms_hip_graph_exchanges); "
                         "auto = the library's default (off: measured slower in wall time on this stack)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the extra leg of the default run: BENCHMARK3_MPDATA (BASELINE.json configuration 5) timed "
                         "after the headline configuration and reported under the key config5")
    ap.add_argument("--no-physics", dest="physics", action="store_false",
                    help="keep the outputs of bulk_flux + set_vbc fixed instead of recomputing them on the "
                         "device every step (SURVEY 8f-1); default: recompute, as the reference's step does")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from roms_trunk_mgh_amd import hip, main3d

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.gpus not in TILINGS:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    ntI, ntJ = TILINGS[args.gpus]

    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: the HIP path has no CPU fallback")
    # one rank per GPU; a rehearsal of N ranks on a one-GPU box (ROMS_BENCH_SHARE_GPU=1) folds them
    device = local_rank % torch.cuda.device_count() if os.environ.get("ROMS_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(device)

    # N > 1: a rank that dies leaves the others waiting in a collective for ever (ncclCommInitRank, a gloo
    # all-reduce, a halo exchange) -- give up loudly instead.  Started BEFORE the first collective.
    import threading
    progress = {"t": time.time(), "limit": 120.0, "phase": "set-up"}

    def tick(phase=None, limit=None):
        progress["t"] = time.time()
        if phase:
            progress["phase"] = phase
        if limit:
            progress["limit"] = limit

    def watchdog():
        while True:
            time.sleep(2.0)
            if time.time() - progress["t"] > progress["limit"]:
                sys.stderr.write(f"[bench rank {rank}] no progress for {progress['limit']:.0f} s in phase "
                                 f"'{progress['phase']}' (a peer rank failed?): aborting\n")
                sys.stderr.flush()
                os._exit(3)
    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    # RCCL prints a version banner on stdout when it is first used; this file's stdout carries
    # exactly one JSON line, so stdout points at stderr while the transport is set up and warmed up
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    uid = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, max-reduce of timings, id broadcast): gloo on the host;
        # data plane (halo swaps): RCCL inside libroms_hip.so
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        tick("unique id")
        import ctypes
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            lib = hip.load()
            if lib.roms_hip_get_unique_id(buf) != 0:
                raise SystemExit("ncclGetUniqueId failed")
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        uid = bytes(t.numpy().tobytes())

    def barrier():
        if world > 1:
            dist.barrier()

    def all_min(flag):
        if world == 1:
            return flag
        t_ = torch.tensor([flag], dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.MIN)
        return int(t_.item())

    tick("make_tile", 300.0)
    st = make_tile(args.config, ntileI=ntI, ntileJ=ntJ, tile=rank, perturb=1.0)
    st.p.mpdata_fast = 0 if args.mpdata_exact else 1
    # halo transport: RCCL (ncclSend/ncclRecv inside the library) unless ROMS_BENCH_HALO=relay, or the
    # communicator cannot be created on every rank, or the halo self-test fails under it; the relay moves the
    # same packed ghost lines through pinned host memory + gloo (slower; recorded in config.halo_transport)
    transport = "none" if world == 1 else "rccl"
    be = None
    selftest = None
    notes = []

    if world > 1 and os.environ.get("ROMS_BENCH_HALO") != "relay":
        tick("ncclCommInitRank", 120.0)
        ok_init = 1
        try:
            be = hip.RomsHip(st, rank=rank, device=device, nccl_unique_id=uid)
        except RuntimeError as e:
            sys.stderr.write(f"[bench rank {rank}] RCCL transport unavailable: {e}\n")
            ok_init = 0
        if all_min(ok_init) == 0:
            notes.append("ncclCommInitRank failed on at least one rank")
            if be is not None:
                be.close()
            be = None
            transport = "relay"
        else:
            tick("halo self-test (rccl)", 120.0)
            try:
                bad = halo_selftest(be, st, rank)
            except RuntimeError as e:
                sys.stderr.write(f"[bench rank {rank}] halo self-test under RCCL raised: {e}\n")
                bad = -1
            if all_min(1 if bad == 0 else 0) == 0:
                notes.append("halo self-test failed under RCCL")
                be.close()
                be = None
                transport = "relay"
            else:
                selftest = "ok"
    elif world > 1:
        transport = "relay"
    if be is None and world == 1 and args.loopback:
        import ctypes
        buf = ctypes.create_string_buffer(128)
        if hip.load().roms_hip_get_unique_id(buf) != 0:
            raise SystemExit("--loopback: RCCL is not available")
        tick("backend (loopback)", 120.0)
        be = hip.RomsHip(st, rank=0, device=device, nccl_unique_id=bytes(buf.raw))
        transport = "rccl-loopback"
    if be is None:
        tick("backend", 120.0)
        be = hip.RomsHip(st, rank=rank, device=device, nccl_unique_id=None)
        if transport == "relay":
            be.set_halo_relay_gloo(dist, torch)
            tick("halo self-test (relay)", 120.0)
            bad = halo_selftest(be, st, rank)
            selftest = "ok" if all_min(1 if bad == 0 else 0) == 1 else "FAILED"
    # how many ranks run their halos through an RCCL communicator of `world` ranks
    rccl_ranks = 0
    if world > 1:
        t_ = torch.tensor([1 if transport == "rccl" else 0], dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.SUM)
        rccl_ranks = int(t_.item())
    if args.graph_exchanges != "auto":
        be.graph_exchanges(args.graph_exchanges == "on")
    m = main3d.Main3D(be, physics=args.physics, diagnostics=args.physics)      # NINFO == 1 (roms_benchmark3.in:257)
    if world > 1 and args.physics:
        # diag.F:398-420: the tile-local results are reduced over the ranks every time (mp_reduce / mp_reduce2)
        tile_diag = be.diag

        def global_diag(s_):
            v = torch.from_numpy(tile_diag(s_))
            allv = [torch.zeros(12, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allv, v)
            return main3d.reduce_diag(torch.stack(allv).numpy())
        be.diag = global_diag
    tick("warm-up", 180.0)
    m.initial()
    for _ in range(args.warmup):
        m.step()
        tick()
    be.sync()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    be.sync()
    torch.cuda.synchronize()
    barrier()
    tick("timed region", 180.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.step()
        tick()
    be.sync()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    tick("kernel timing")
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    # ---- roofline of the dominant graded kernel: step3d_t (live hipEvent timing) ----
    be.timing(True)
    per_kernel = {}
    names = ["ana_srflux", "bulk_flux", "set_vbc", "lmd_vmix", "wvelocity", "diag", "set_massflux", "rho_eos", "omega", "set_zeta", "pre_step3d", "prsgrd", "t3dmix2", "rhs3d_tile",
             "uv3dmix2", "step2d_loop", "set_depth", "step3d_uv", "step3d_t"]
    acc = {n: [] for n in names}
    for _ in range(5):
        m.step()
        tick()
        for n in names:
            v = be.last_ms(n)
            if v >= 0:
                acc[n].append(v)
    for n in names:
        if acc[n]:
            per_kernel[n] = sum(acc[n]) / len(acc[n])
    b = st.b
    # achievable bandwidth next to the vendor peak: a streaming copy in the library's access
    # pattern, 8 B read + 8 B written per double, on a working set of one 3-D field
    ncal = (b.UBi - b.LBi + 1) * (b.UBj - b.LBj + 1) * b.N
    cal = []
    for _ in range(7):
        be.calib_stream(ncal)
        cal.append(be.last_ms("calib_stream"))
    cal.sort()
    measured_peak = 16.0 * ncal / (cal[len(cal) // 2] * 1e-3) / 1e9
    be.timing(False)
    graph_state = be.graph_exchanges_state()      # 1: LOOP_2D ran as one hipGraph with its exchanges inside (loopback / opt-in)
    tile_cells = (b.Iend - b.Istr + 1) * (b.Jend - b.Jstr + 1) * b.N
    alg_bytes = 8.0 * (4 * b.NT + 4) * tile_cells          # SURVEY.md section 8d (compulsory traffic)
    t_ms = per_kernel.get("step3d_t", float("nan"))
    achieved = alg_bytes / (t_ms * 1e-3) / 1e9
    mpdata = any(st.p.Hadv[i] == 6 for i in range(b.NT))   # enum roms_adv: ADV_MPDATA
    # finite check: the timed run must not have blown up
    be.to_host(["zeta", "t"])
    import numpy as np
    ok = bool(np.isfinite(st["zeta"]).all() and np.isfinite(st["t"]).all())
    be.close()

    if rank == 0:
        dt = st.p.dt
        value = args.steps * dt / 86400.0 / wall
        adv = ("MPDATA tracer advection (all tracers; mpdata_adiff quotients: "
               + ("IEEE divisions)" if args.mpdata_exact else "refined reciprocals, 1e-10 relRMS of the exact kernel)")
               if mpdata else "U3/C4 tracer advection")
        cfg = {"workload": f"{args.config} {b.Lm}x{b.Mm}x{b.N} NT={b.NT} "
                           f"nonlinear 3-D step incl. {2 * st.p.nfast + 1} step2d calls, {adv}, "
                           + ("analytic atmospheric forcing, bulk fluxes + KPP + diagnostics every step"
                              if args.physics else "fixed forcing / mixing fields"),
               "tiling": f"{ntI}x{ntJ}", "halo_transport": transport, "rccl_ranks": rccl_ranks,
               "halo_selftest": selftest, "graph_exchanges": graph_state,
               "per_step_physics": "ana_srflux+bulk_flux+set_vbc+lmd_vmix (KPP)+wvelocity+diag (NINFO=1) on device"
               if args.physics else "fixed inputs", "dt_s": dt, "ndtfast": st.p.ndtfast, "finite": ok}
        if notes:
            cfg["notes"] = notes
        roof = {"kernel": ("k_mp_ta + k_mp_adiff + k_mp_update per tracer (step3d_t_tile, MPDATA)" if mpdata
                           else "k_step3d_t_pipe (step3d_t_tile)"),
                "bound": "hbm", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.config, args.gpus)[0], "traffic_unit": "bytes/launch",
                "traffic_source": pmc_traffic(args.config, args.gpus)[1],
                "avg_ms": t_ms, "algorithmic_bytes": alg_bytes,
                "measured_peak": measured_peak,
                "measured_peak_note": "streaming copy k_calib_stream, 8 B read + 8 B written per double, one 3-D field"}
        if mpdata:
            multi = 8.0 * (12 * b.NT + 6) * tile_cells     # Ta, Ua, Va, Wa, beta_up/dn materialised
            roof["multipass_bytes"] = multi
            roof["multipass_frac"] = multi / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "simulated-days/wall-sec", "value": value, "unit": "simulated-days/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": cfg, "roofline": roof, "kernel_ms": per_kernel,
        }
        if args.gpus == 1 and args.config == "BENCHMARK3" and not args.no_config5 and not args.loopback:
            # BASELINE.json configuration 5 on the same box, after the headline run (its own state and context)
            try:
                out["config5"] = run_config5(args, device)
            except Exception as e:            # the headline line must not depend on this leg
                out["config5"] = {"error": repr(e)}
        if not args.no_cpu_baseline and args.gpus == 1:
            # the GPU boxes of this pool report every core of the host (256) but give a one-GPU job a share of
            # 16 (their process guard and memory cap are sized for that): the baseline uses that share
            ncores = args.cpu_cores or min(len(os.sched_getaffinity(0)), CPU_SHARE)
            out["cpu_baseline"] = cpu_baseline(args.config, args.physics, args.cpu_steps, ncores)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
