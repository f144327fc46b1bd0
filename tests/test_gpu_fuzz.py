"""Seed-pinned subset of the randomised parity sweeps (scripts/fuzz_parity.py, scripts/fuzz_lockstep.py) in the GPU tier — VERDICT r1 item 9.
Every configuration is bit for bit against the CPU oracle: states, all counters, event times / root stops, failure counts."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture
def det_pow(O):
    O.set_det_pow(True)
    yield
    O.set_det_pow(False)


@pytest.mark.parametrize("seed", range(14))
def test_device_resident_integrators_random_configuration(H, O, det_pow, seed):
    """7 models x BDF / TR-BDF2 / ESDIRK34 x per-member / wavefront lock-step x rtol 1e-9..1e-3 x random parameter ranges, events, DAEs, banded
    run-time-sized models, runs that fail: seeds 1000..1013 (each model twice)."""
    ok, msg = fuzz_cases.resident_case(H, O, seed)
    assert ok, msg


@pytest.mark.parametrize("seed", range(18))
def test_host_driven_lockstep_integrators_random_configuration(H, O, seed):
    """9 models x three methods x nbatch 1..8192 x fused / trait-only x rtol 1e-8..1e-3: seeds 2000..2017 (each model twice, four 8192-member runs)."""
    ok, msg = fuzz_cases.lockstep_case(H, O, seed)
    assert ok, msg


def test_pinned_sweep_of_the_wavefront_per_member_kernels():
    """scripts/fuzz_wave_member.py, 12 seed-pinned configurations (48 ran clean: profiles/r02_fuzz_wave_member.txt): BDF / TR-BDF2 / ESDIRK34 in the
    wavefront-per-member form on run-time-sized built-in models of random sizes, tolerances and parameters, events included, bit for bit against
    independent oracle solves.  In a child process: the script forces the banded models off their lane-per-member twins (DSH_RESIDENT_LANE=0)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_wave_member.py"), "12", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "12 of 12 configurations bit-identical" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

