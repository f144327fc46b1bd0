"""Randomised parity sweep of the device-resident integrators against the CPU oracle (bit for bit, deterministic pow); the configurations are
tests/fuzz_cases.py::resident_case (a seed-pinned subset runs in the GPU test tier).  python scripts/fuzz_parity.py [nseeds]   (needs a GPU; FUZZ_BASE = first seed)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from fuzz_cases import resident_case  # noqa: E402

O.build()
O.set_det_pow(True)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for seed in range(nseeds):
    ok, msg = resident_case(H, O, seed, base=int(os.environ.get("FUZZ_BASE", "1000")))
    bad += 0 if ok else 1
    print(msg, flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
