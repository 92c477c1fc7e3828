"""examples/lqg2d_pi.c: a plain C program with the reference's call sequence (tprob_test.c:2275-2364) linked against
libc3sc.so -- set-up, init_value, pi_solve / vi_solve, save / load, closed-loop simulation -- with both control
minimisers (candidate list, box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "c3sc_amd", "host")


def _build(tmp_path):
    exe = str(tmp_path / "lqg2d_pi")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "lqg2d_pi.c"), "-L", HOST, "-lc3sc", "-lm",
                           f"-Wl,-rpath,{HOST}", f"-Wl,-rpath,{os.path.join(ROOT, 'c3sc_amd', 'csrc')}", "-o", exe])
    return exe


def test_example_compiles_against_the_public_headers(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("minimiser", ["bruteforce", "bfgs"])
def test_example_runs_end_to_end(tmp_path, minimiser):
    exe = _build(tmp_path)
    # bruteforce at the reference regression's own size (100 x 100, maxrank 20: the rank-20 core of a 100-node dimension
    # does not fit LDS and takes the unstaged kernel), the box minimiser on a smaller grid
    n = "100" if minimiser == "bruteforce" else "40"
    p = subprocess.run([exe, n, "3", "6.0", minimiser], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    print(p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0 and "LQG2D_PI_OK" in p.stdout
