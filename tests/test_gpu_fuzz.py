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
