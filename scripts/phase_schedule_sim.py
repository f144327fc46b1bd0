#!/usr/bin/env python
"""Why in-wavefront scheduling of divergent phases is bounded: a Monte-Carlo model of one 64-lane wavefront of the per-member BDF on the C2
phase statistics (per member ~264 steps, ~2 Newton iterations per step, a step-size change / refactorisation on ~23 % of the steps, a few rejected steps).
Phase costs in wave instructions from the PMC counts of the lock-step kernel: NEWTON 380, ERRTEST (+ order selection) 660, RESCALE (R U + LU) 1900.
Policies: `all` = every phase with a taker runs in every pass (what nested per-lane loops amount to), `count` = the most populated phase only
(k_bdf_member_sched), `pow a` = argmax count / cost^a.  `ideal` = the slowest single lane (no scheduler inside one wavefront can beat it).
    python scripts/phase_schedule_sim.py"""
import numpy as np

COST = {"N": 380, "E": 660, "R": 1900}


def member_seq(rng):
    seq = []
    for _ in range(int(rng.normal(264, 25))):
        k = rng.choice([1, 2, 3, 4], p=[0.25, 0.55, 0.15, 0.05])
        if rng.random() < 0.02:
            seq += ["N"] * 3 + ["R"]
        seq += ["N"] * k + ["E"]
        if rng.random() < 0.008:
            seq += ["R"] + ["N"] * 2 + ["E"]
        if rng.random() < 0.23:
            seq += ["R"]
    return seq


def simulate(policy, seed, nl=64):
    rng = np.random.default_rng(seed)
    seqs = [member_seq(rng) for _ in range(nl)]
    pos = [0] * nl
    total = 0
    while True:
        cnt = {"N": 0, "E": 0, "R": 0}
        for i in range(nl):
            if pos[i] < len(seqs[i]):
                cnt[seqs[i][pos[i]]] += 1
        if sum(cnt.values()) == 0:
            break
        picks = [ph for ph in "NER" if cnt[ph]] if policy == "all" else [policy(cnt)]
        for pick in picks:
            total += COST[pick]
            for i in range(nl):
                if pos[i] < len(seqs[i]) and seqs[i][pos[i]] == pick:
                    pos[i] += 1
    return total, max(sum(COST[p] for p in s) for s in seqs), float(np.mean([sum(COST[p] for p in s) for s in seqs]))


if __name__ == "__main__":
    pols = {"all": "all", "count": lambda c: max("NER", key=lambda p: (c[p], -COST[p]))}
    for a in (0.25, 0.5, 0.75):
        pols[f"pow {a}"] = (lambda a: lambda c: max("NER", key=lambda p: c[p] / COST[p] ** a))(a)
    for name, pol in pols.items():
        r = [simulate(pol, s) for s in range(4)]
        print(f"{name:9s} wave instructions {np.mean([x[0] for x in r]) / 1e3:7.0f} k   slowest lane {np.mean([x[1] for x in r]) / 1e3:5.0f} k   mean lane {np.mean([x[2] for x in r]) / 1e3:5.0f} k")
