"""A plain-C program built with gcc against include/snappyhash.h and libsnappyhash.so: the
header is valid C and the ABI needs nothing but pointers and sizes."""
import os
import subprocess

import pytest

import helpers as H

SRC = os.path.join(H.ROOT, "tests", "c_driver", "driver.c")


def _build(tmp_path):
    import torch  # the library links the HIP runtime that torch ships: put it on the loader path

    import snappy_amd  # noqa: F401  (makes sure the library is built)

    exe = str(tmp_path / "driver")
    libdir = os.path.join(H.ROOT, "snappy_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(H.ROOT, "include"),
           SRC, "-o", exe, "-L", libdir, "-lsnappyhash", "-Wl,-rpath," + libdir,
           "-Wl,--allow-shlib-undefined"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    env = dict(os.environ)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env["LD_LIBRARY_PATH"] = torch_lib + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return exe, env


def test_c_driver_host_entry_points(tmp_path):
    exe, env = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c driver ok (host)" in out.stdout


@pytest.mark.gpu
def test_c_driver_on_gpu(tmp_path):
    exe, env = _build(tmp_path)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c driver ok (gpu)" in out.stdout
