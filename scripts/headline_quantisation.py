#!/usr/bin/env python
"""Wave quantisation of the headline kernel (k_bdf_adaptive, wavefront lock-step groups of 64): 100 000 members are 1563 wavefronts on 1024 SIMDs — 539 SIMDs
hold two wavefronts, 485 hold one, and the launch lasts as long as the doubly occupied ones.  Times the same solve at ensemble sizes that give every SIMD exactly
one wavefront (65 536 members), the bench's 100 000, exactly two (131 072) and more (HIP events around the launch, the library's kernel timing):
    python scripts/headline_quantisation.py        (GPU only)
Per size: kernel ms, ns per member, member-steps/s.  T(65 536) is one wavefront's dependent chain (latency bound), T(131 072) two sharing a SIMD."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

print("| members | wavefronts | per SIMD | kernel ms | ns / member | member-steps / s |")
print("|---|---|---|---|---|---|")
for nb in (16384, 32768, 65536, 100000, 131072, 196608, 262144, 400000):
    p = robertson_params(max(nb, 100000))[:nb]
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, block_threads=256)
    for _ in range(2):
        s.solve_dense(T_EVAL, want_host=False)
    s.set_kernel_timing(True)
    steps = 0
    for _ in range(5):
        s.solve_dense(T_EVAL, want_host=False)
        steps = s.last_solve_info()[1]["number_of_steps"]
    nl, ms = s.kernel_timing()
    s.set_kernel_timing(False)
    t = ms / nl
    w = (nb + 63) // 64
    print(f"| {nb} | {w} | {w / 1024:.2f} | {t:.3f} | {1e6 * t / nb:.2f} | {steps / (t * 1e-3):.3e} |", flush=True)
    del s
