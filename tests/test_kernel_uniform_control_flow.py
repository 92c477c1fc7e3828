"""The fiber-pair kernels compute wave-uniform tables lane-distributed (NodeRegs, CandRegs: v_readlane) before parking
them in LDS.  A register the compiler spills and reloads inside a lane-divergent branch is only restored
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
                                        ("inst_lqg6_fpp.hip", "k_fiber_pair")])
def test_no_lane_divergent_control_flow(tmp_path, src, kernel):
    out = tmp_path / "k.s"
    subprocess.run([HIPCC, "-std=c++20", "-O3", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)], check=True, cwd=CSRC,
                   stderr=subprocess.DEVNULL)
    name, seen = None, 0
    body = []
    for line in open(out):
        m = re.match(r"^(_ZN4c3sc\d+%s\S*):" % kernel, line)
        if m:
            name, body = m.group(1), []
        elif name and line.startswith(".Lfunc_end"):
            # the only EXEC writes allowed belong to the epilogue `if (st) atomicOr(status, st)`: a saveexec + a
            # single-lane mask right before global_atomic_or / s_endpgm (the block may be tail-duplicated and placed
            # anywhere in the listing)
            bad = []
            for i, l in enumerate(body):
                if EXEC_WRITE.search(l) and not any("global_atomic_or" in t or "s_endpgm" in t for t in body[i:i + 24]):
                    bad.append(i)
            assert not bad, f"{name}: {len(bad)} writes to EXEC outside the epilogue -- lane-divergent control flow crept in"
            seen += 1
            name = None
        elif name:
            body.append(line)
    assert seen > 0
