"""Collect the compiler's per-kernel resource usage (-Rpass-analysis=kernel-resource-usage remarks the Makefile saves next to
every object as *.res) into c3sc_amd/csrc/kernel_resources.json: registers, spilled registers, scratch bytes per lane,
waves per SIMD.  bench.py prints the entry of the kernel it timed.
    python tools/kernel_resources.py c3sc_amd/csrc"""
import glob
import json
import os
import re
import subprocess
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "c3sc_amd", "csrc")
out, cur = {}, None
for fn in sorted(glob.glob(os.path.join(d, "*.res"))):
    for line in open(fn, errors="replace"):
        m = re.search(r"remark: (?:Function Name: (\S+)|\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = out.setdefault(m.group(1), {})
        elif cur is not None:
            key = {"VGPRs": "vgprs", "AGPRs": "agprs", "SGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
                   "Occupancy [waves/SIMD]": "waves_per_simd", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill",
                   "LDS Size [bytes/block]": "static_lds_bytes"}[m.group(2)]
            cur[key] = int(m.group(3))
names = list(out)
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
except (OSError, subprocess.CalledProcessError):
    dem = names
res = {}
for n, dn in zip(names, dem):
    dn = re.sub(r"^void ", "", dn)
    dn = re.sub(r"\(c3sc::KArgs.*$", "", dn).replace("c3sc::", "")
    res[dn] = out[n]
json.dump(res, open(os.path.join(d, "kernel_resources.json"), "w"), indent=0, sort_keys=True)
print(f"{len(res)} kernels -> {os.path.join(d, 'kernel_resources.json')}")
