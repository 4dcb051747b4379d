"""GPU: the RCCL leg of the halo exchange (halo.hip: ncclCommInitRank, one ncclGroupStart/End with the
ncclSend/ncclRecv of an exchange) executed on hardware.

* loopback (runs on the one-GPU test box): ONE tile initialised WITH an RCCL id -- the tile is its own W and
  E neighbour, its periodic wrap goes out and comes back through RCCL (two sends to and two receives from
  the same peer per exchange, paired by order: the assumption the 2x1 and 2x2 periodic tilings rely on) and
  every kernel takes its multi-tile branch incl. the deferred-flux step2d call.  Result = the plain
  one-tile run, bit for bit.
* 2x1 on two devices: skipped with an explicit reason when the box has one GPU (RCCL refuses two ranks on
  one device); otherwise one rank per device through RCCL, bit-equal to the one-tile run.
mp_exchange2d/3d/4d: ROMS/Utility/mp_exchange.F:73-286, 290-902."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_multitile import HERE, _free_port, _single

pytestmark = pytest.mark.gpu


def _run_rccl(tmp_path, world, ntI, ntJ, config, nsteps, variant):
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_gpu_worker.py"), str(r), str(world), str(ntI),
                               str(ntJ), config, str(nsteps), str(port), str(tmp_path), variant], env=env)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def _compare(tmp_path, ref, world):
    rb = ref.b
    for r in range(world):
        d = np.load(os.path.join(tmp_path, f"tile{r}.npz"))
        Istr, Iend, Jstr, Jend, LBi, LBj = [int(x) for x in d["bounds"]]
        for name in ("zeta", "ubar", "vbar", "u", "v", "t", "Huon", "W", "Hz"):
            a = d[name]
            ni, nj = a.shape[0], a.shape[1]
            i0, j0 = LBi - rb.LBi, LBj - rb.LBj
            want = ref[name][i0:i0 + ni, j0:j0 + nj]
            own = (slice(Istr - LBi, Iend - LBi + 1), slice(Jstr - LBj, Jend - LBj + 1))
            assert np.array_equal(a[own], want[own]), (name, r, float(np.abs(a[own] - want[own]).max()))
            if name in ("zeta", "t", "Hz", "W"):      # rho-type: every ghost point is defined
                iv = min(ni, rb.Lm + rb.NghostPoints - LBi + 1)
                jv = min(nj, rb.Mm + 1 - LBj + 1)
                assert np.array_equal(a[:iv, :jv], want[:iv, :jv]), (name, r, "ghost points differ")


@pytest.mark.parametrize("config,variant", [("BENCHMARK_TINY", "physics+rccl"), ("UPWELLING", "rccl"),
                                            ("BENCHMARK_TINY", "mpdata+rccl"), ("SEAMOUNT", "mask+rccl")])
def test_rccl_loopback_equals_local_periodic_copy(tmp_path, config, variant):
    nsteps = 3
    ref, _ = _single(config, nsteps, variant.replace("+rccl", "").replace("rccl", ""))
    _run_rccl(tmp_path, 1, 1, 1, config, nsteps, variant)
    _compare(tmp_path, ref, 1)


def test_rccl_two_devices(tmp_path):
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"needs 2 GPUs for one RCCL rank per device (this box has {ndev}); the transport itself "
                    "runs in the loopback test above")
    nsteps = 3
    ref, _ = _single("BENCHMARK_TINY", nsteps, "physics")
    _run_rccl(tmp_path, 2, 2, 1, "BENCHMARK_TINY", nsteps, "physics+rccl")
    _compare(tmp_path, ref, 2)
