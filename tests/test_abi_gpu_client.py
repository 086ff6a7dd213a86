"""The drop-in boundary from a host with no Python and no torch in it, ON THE GPU ("the only non-Python client still launches nothing"):
gcc compiles tests/c/abi_gpu_client.c against include/e2eft.h alone; the program dlopens libamdhip64.so for device memory and libe2eft.so for the
kernels, launches e2eft_groupnorm_fwd and e2eft_conv2d_fwd on buffers it owns and checks three known answers (see the C file)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_launches_kernels(dev, tmp_path):
    from diffusion_e2e_ft_amd import _lib
    _lib.load()
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    exe = str(tmp_path / "abi_gpu_client")
    r = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_gpu_client.c"),
                        "-o", exe, "-ldl", "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hip = next((p for p in ("/opt/rocm/lib/libamdhip64.so", "/opt/rocm/lib64/libamdhip64.so") if os.path.exists(p)), "libamdhip64.so")
    r = subprocess.run([exe, _lib.lib_path(), hip], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "from plain C" in r.stdout
