"""Checks on the generated gfx950 code that no run-time test can give.

The lane-grid K1 (`csrc/egp_pd_grid.hpp`) issues `v_fmac_f64_dpp` / `v_mov_b64_dpp` from inline asm. A DPP read of a VGPR
needs two wait states after a VALU write of it, and the compiler's hazard recogniser does not look into inline asm: the
kernel relies on the order of its asm statements (and on `s_nop` where a source is fresh). Whether that holds is a
property of the listing the compiler produced, so the listing is checked (tools/isa_stats.py) -- a wrong schedule would
read stale registers only on some inputs and some compiler versions."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_grid_kernel_listing_has_no_dpp_hazard_and_no_scratch(tmp_path):
    asm = tmp_path / "k.s"
    src = os.path.join(ROOT, "egopose_amd", "csrc", "egp_kernels.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(asm), src],
                   check=True, capture_output=True, timeout=600)
    for kernel in ("k_pd_torque_grid58IdE", "k_pd_torque_grid58IfE"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), str(asm), kernel],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "DPP hazards: 0" in r.stdout
        assert " 0 scratch" in r.stdout, r.stdout            # 15 x 4 doubles of matrix per lane must stay in registers
        assert "v_fmac_f64_dpp" in r.stdout
