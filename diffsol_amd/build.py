"""Build the native libraries of diffsol_amd in-tree (diffsol_amd/lib/*.so) for gfx950.

    python -m diffsol_amd.build            # device library (hipcc) + host library (g++)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container; the resulting .so files travel
with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=8):
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"], check=True, stdout=out)
    subprocess.run(["make", "-C", os.path.join(_HERE, "host")], check=True, stdout=out)
    return [os.path.join(_HERE, "lib", "libdiffsol_hip.so"), os.path.join(_HERE, "lib", "libdiffsol_hip_host.so")]


if __name__ == "__main__":
    for p in build(verbose="-v" in sys.argv):
        print(p)
