"""The drop-in boundary from plain C: include/mm3d.h must be valid C99 and C++11, and a C program must
link against libmm3d.so with nothing but gcc (examples/mm3d_demo.c: what a cgo / JNI / N-API stub does)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "map-merge_amd")


def _run(cmd, **kw):
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, **kw)


def test_header_is_plain_c99_and_cxx11():
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-Iinclude", "-x", "c", "include/mm3d.h"],
                ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-Iinclude", "-x", "c++", "include/mm3d.h"]):
        r = _run(cmd)
        assert r.returncode == 0, r.stderr


def _build_demo(tmp_path):
    exe = str(tmp_path / "mm3d_demo")
    r = _run(["gcc", "-std=c99", "-D_DEFAULT_SOURCE", "-O2", "-Wall", "-Wextra", "-Werror", "-Iinclude", "examples/mm3d_demo.c", "-L" + LIBDIR, "-lmm3d",
              "-Wl,-rpath," + LIBDIR, "-lm", "-o", exe])
    assert r.returncode == 0, r.stderr
    return exe


def test_c_program_links_against_the_library(tmp_path, mm):
    _build_demo(tmp_path)            # (mm: the library has been built)


@pytest.mark.gpu
def test_c_program_runs(tmp_path, mm):
    exe = _build_demo(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pairs estimated: 1" in r.stdout and "translation error" in r.stdout
    err = float(r.stdout.split("translation error")[1].split()[0])
    assert err < 0.2, r.stdout          # matching + RANSAC + ICP recovers the pose of this scene
    # the same program on a device list of one: plain C through mm3d_create_devices, the records through ncclAllGather
    import os
    r2 = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, MM3D_DEMO_DEVICES="1"))
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "device list of 1, pair records through ncclAllGather" in r2.stdout
    line = lambda out: [l for l in out.splitlines() if l.startswith("recovered yaw")][0]     # noqa: E731
    assert line(r2.stdout) == line(r.stdout)
