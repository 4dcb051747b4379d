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
