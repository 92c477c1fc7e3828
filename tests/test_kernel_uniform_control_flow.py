"""The fiber-pair / fiber-per-lane kernels keep wave-uniform tables in VGPR lanes (NodeRegs, CandRegs) and read
them with v_readlane.  A register the compiler spills and reloads inside a lane-divergent branch is only restored
for the active lanes, so those kernels must not contain lane-divergent control flow at all (kernel_common.hpp,
node_backup).  This test compiles the headline instantiations to ISA and checks that EXEC is only touched by the
single-lane status atomic of the epilogue."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "c3sc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXEC_WRITE = re.compile(r"saveexec|s_mov_b64 exec|s_and_b64 exec|s_andn2_b64 exec|s_xor_b64 exec")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,kernel", [("inst_car7d_fpp.hip", "k_fiber_pair"), ("inst_other_fpp.hip", "k_fiber_pair"),
                                        ("inst_lqg6_fpp.hip", "k_fiber_pair"), ("inst_car7d_fpl.hip", "k_fiber_per_lane")])
def test_no_lane_divergent_control_flow(tmp_path, src, kernel):
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "-std=c++20", "-O3", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, cwd=CSRC,
                   stderr=subprocess.DEVNULL)
    name, sites, seen = None, 0, 0
    for line in open(out):
        m = re.match(r"^(_ZN4c3sc\d+%s\S*):" % kernel, line)
        if m:
            name, sites = m.group(1), 0
        elif name and EXEC_WRITE.search(line):
            sites += 1
        elif name and line.startswith(".Lfunc_end"):
            # epilogue: `if (st) atomicOr(status, st)` = one saveexec + one single-lane mask
            assert sites <= 2, f"{name}: {sites} writes to EXEC -- lane-divergent control flow crept in"
            seen += 1
            name = None
    assert seen > 0
