"""CPU: the ISO_C_BINDING shim module compiles with flang (when present) and its
field-id constants agree with include/roms_fields.def."""
import os
import re
import shutil
import subprocess

import pytest

from roms_trunk_mgh_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "roms_trunk_mgh_amd", "fortran", "roms_hip_mod.F90")


def test_field_ids_match_def_file():
    pairs = re.findall(r"FID_(\w+)=(\d+)", open(SRC).read())
    assert len(pairs) == len(abi.FIELDS)
    assert all(abi.FIELD_ID[n] == int(v) for n, v in pairs)


@pytest.mark.skipif(shutil.which("flang") is None, reason="flang not installed")
def test_shim_compiles(tmp_path):
    r = subprocess.run(["flang", "-c", SRC, "-o", str(tmp_path / "m.o"), "-module-dir", str(tmp_path)],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(shutil.which("flang") is None, reason="flang not installed")
def test_fortran_struct_mirrors_have_the_abi_sizes(tmp_path):
    """TYPE(roms_bounds_t), TYPE(roms_params_t), TYPE(roms_step_idx_t), TYPE(roms_halo_msg_t) of roms_hip_mod
    occupy exactly the bytes the C structs do (roms_abi_sizeof / ctypes mirror)."""
    import ctypes
    from roms_trunk_mgh_amd import hip
    prog = tmp_path / "sz.F90"
    prog.write_text("""program sz
  use, intrinsic :: iso_c_binding
  use roms_hip_mod
  type(roms_bounds_t) :: b
  type(roms_params_t) :: p
  type(roms_step_idx_t) :: s
  type(roms_halo_msg_t) :: m
  print '(4(i0,1x))', c_sizeof(b), c_sizeof(p), c_sizeof(s), c_sizeof(m)
end program
""")
    r = subprocess.run(["flang", "-c", SRC, "-o", "m.o"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["flang", str(prog), "m.o", "-o", "sz"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(tmp_path / "sz")], capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [ctypes.sizeof(abi.Bounds), ctypes.sizeof(abi.Params), ctypes.sizeof(abi.StepIdx),
                                     ctypes.sizeof(hip.HaloMsg)]
