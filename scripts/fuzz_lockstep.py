"""Randomised parity sweep of the HOST-DRIVEN lock-step integrators against the CPU oracle's lock-step batched run (libm pow on both sides), bit for bit; the
configurations are tests/fuzz_cases.py::lockstep_case (a seed-pinned subset runs in the GPU test tier).  python scripts/fuzz_lockstep.py [nseeds]   (needs a GPU)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from fuzz_cases import lockstep_case  # noqa: E402

O.build()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for seed in range(nseeds):
    ok, msg = lockstep_case(H, O, seed, base=int(os.environ.get("FUZZ_BASE", "2000")))
    bad += 0 if ok else 1
    print(msg, flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
