#!/usr/bin/env python3
"""Per-kernel resource metadata (VGPRs, SGPRs, scratch bytes, LDS) of the gfx950 code object embedded in a hipcc
object file or shared library: `tools/kernel_meta.py a.o [b.o]` prints one line per kernel, or the kernels whose
numbers differ between the two files."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def meta(path):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out[dem] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                        lds=g("group_segment_fixed_size"))
    return out


def main():
    a = meta(sys.argv[1])
    if len(sys.argv) == 2:
        for k, v in sorted(a.items()):
            print(f"{v['vgpr']:4d} vgpr {v['sgpr']:4d} sgpr {v['scratch']:5d} scratch {v['lds']:6d} lds  {k}")
        return
    b = meta(sys.argv[2])
    n = 0
    for k in sorted(set(a) | set(b)):
        if a.get(k) != b.get(k):
            n += 1
            print(f"{k}\n    {a.get(k)}\n -> {b.get(k)}")
    print(f"{n} of {len(set(a) | set(b))} kernels differ")


if __name__ == "__main__":
    main()
