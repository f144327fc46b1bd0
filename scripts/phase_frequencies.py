#!/usr/bin/env python
"""How often does the lock-step BDF of the headline workload take each of its paths?  Runs the CPU oracle (bit-identical to k_bdf_adaptive<.., WAVE> on a
64-member group in deterministic-pow mode) on a few wavefront-sized groups of the BASELINE config-2 ensemble and prints events per step — the weights of
the per-phase instruction account in profiles/r03_isa_account.md.   python scripts/phase_frequencies.py [groups]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import oracle as O
from bench import robertson_params as ensemble_parameters  # the same synthetic ensemble as bench.py

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4
O.build()
O.set_det_pow(True)
p_all = ensemble_parameters(100000)
tot = {}
stats = np.zeros(5)
for g in range(groups):
    p = p_all[g * 64 * 97 % (100000 - 64):][:64]
    o = O.OracleSolver(O.MODEL_ROBERTSON_ODE, p, nbatch=64, model_size=1, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    O.event_counts(reset=True)
    o.solve(4e5)
    ev = O.event_counts(reset=True)
    st = o.stats()
    for k, v in ev.items(): tot[k] = tot.get(k, 0) + v
    stats += np.array([st["number_of_steps"], st["number_of_nonlinear_solver_iterations"], st["number_of_linear_solver_setups"], st["number_of_error_test_failures"], st["number_of_nonlinear_solver_fails"]])
steps = stats[0]
print(f"{groups} groups of 64: steps {steps:.0f}, Newton iterations {stats[1]:.0f} ({stats[1]/steps:.3f} per step), LU setups {stats[2]:.0f} ({stats[2]/steps:.3f}), error-test failures {stats[3]:.0f}, Newton failures {stats[4]:.0f}")
for k, v in tot.items(): print(f"  {k:32s} {v:8d}   {v/steps:.3f} per step")
