#!/usr/bin/env python3
"""Build-time resource check: fail if any kernel of the product library needs scratch memory or spills registers.
Reads the AMDGPU metadata (.private_segment_fixed_size, .vgpr_spill_count, .sgpr_spill_count) from the device assembly
that `hipcc -save-temps=obj` leaves next to each object (build/*-hip-amdgcn-amd-amdhsa-gfx950.s)."""
import glob
import os
import re
import sys


def kernels(path):
    txt = open(path, errors="replace").read()
    m = re.search(r"amdhsa\.kernels:(.*?)amdhsa\.target", txt, re.S)
    if not m:
        return []
    out = []
    for blk in re.split(r"\n  - ", m.group(1))[1:]:
        def field(name, default=None):
            mm = re.search(r"\." + name + r":\s*(\S+)", blk)
            return mm.group(1) if mm else default
        out.append(dict(name=field("name"), scratch=int(field("private_segment_fixed_size", "0")),
                        vspill=int(field("vgpr_spill_count", "0")), sspill=int(field("sgpr_spill_count", "0")),
                        vgpr=int(field("vgpr_count", "0")), sgpr=int(field("sgpr_count", "0")), lds=int(field("group_segment_fixed_size", "0"))))
    return out


def main(build_dir):
    files = sorted(glob.glob(os.path.join(build_dir, "*gfx950*.s")))
    if not files:
        print(f"check_resources: no device assembly under {build_dir} (build with -save-temps=obj)")
        return 1
    bad, n = [], 0
    for f in files:
        for k in kernels(f):
            n += 1
            if k["scratch"] > 0 or k["vspill"] > 0:
                bad.append((os.path.basename(f), k))
    for f, k in bad:
        print(f"check_resources: {k['name']} ({f}): scratch {k['scratch']} B, {k['vspill']} spilled VGPRs, {k['vgpr']} VGPRs")
    print(f"check_resources: {n} kernels, {len(bad)} with scratch/spills")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "build"))
