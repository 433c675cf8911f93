#!/usr/bin/env python3
"""Registers, spills, scratch, occupancy and LDS of every kernel of one .hip file:
    python tools/kernel_resources.py imagesegmentation_amd/csrc/k_crf_fused_r7.hip
(compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage and condenses the remarks)."""
import re
import subprocess
import sys

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math".split()


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    res = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                         capture_output=True, text=True)
    if res.returncode:
        sys.stderr.write(res.stderr)
        raise SystemExit(res.returncode)
    for block in res.stderr.split("Function Name: ")[1:]:
        name = block.split()[0]
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        name = name.replace("fs::(anonymous namespace)::", "").replace("void ", "").split("(")[0]

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return m.group(1) if m else "?"
        print(f"{name[:70]:70s} V={g('VGPRs'):>3} A={g('AGPRs'):>3} spill={g('VGPRs Spill'):>3} scratch={g('ScratchSize [bytes/lane]'):>5} "
              f"occ={g('Occupancy [waves/SIMD]')} lds={g('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main()
