"""The committed counter summaries bench.py's rooflines read must describe the kernels of THIS tree (a kernel edit two hours after the counters were
taken shipped `roofline.frac: null` for config 5 — the hash guard in bench.py worked, the re-take did not happen).  Every summary carries the sha256 prefix of the
kernel sources it was measured on; this CPU-tier test recomputes the hashes and fails when the NEWEST committed round of any summary is stale, naming the command that
re-takes it (scripts/profile_r06.sh on a GPU box through gpurun, then the publish scripts here).  Older rounds are history and are not checked."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RETAKE = ("kernel sources changed after the counters were taken: on a GPU box `gpurun -- bash scripts/profile_r06.sh`, then here `python scripts/publish_profile.py a r06; "
          "python scripts/publish_profile.py m r06 member; RND=r06 python scripts/publish_configs_profile.py c4_ode:262144 c4_dae:262144 c5_per_member:65536 c5_group64:65536`")


def _load(name):
    assert name, "no committed summary of this kind under profiles/"
    return json.load(open(os.path.join(ROOT, "profiles", name)))


@pytest.mark.parametrize("suffix", ["pmc_resident.json", "pmc_per_member.json"])
def test_the_headline_kernels_counters_are_those_of_this_tree(suffix):
    name = bench.latest_profile(suffix)
    d = _load(name)["bench_kernel"]
    assert d["kernel_source_sha16"] == bench.kernel_source_hash(), f"profiles/{name}: {RETAKE}"
    assert d["members"] == bench.NB_PER_GPU and d["valu_insts_per_launch"] > 0 and d["f64_insts_per_launch"] > 0 and d["hbm_bytes_per_launch"] > 0


@pytest.mark.parametrize("cfg,sources,members", [("c4_ode", "c4", 262144), ("c4_dae", "c4", 262144), ("c5_per_member", "c5", 65536), ("c5_group64", "c5", 65536)])
def test_the_config_kernels_counters_are_those_of_this_tree(cfg, sources, members):
    name = bench.latest_profile("pmc_configs.json")
    d = _load(name).get(cfg)
    assert d, f"profiles/{name} has no entry {cfg}"
    assert d["kernel_source_sha16"] == bench.source_hash(bench.CFG_SOURCES[sources]), f"profiles/{name}[{cfg}]: {RETAKE}"
    assert d["members"] == members and d["valu_insts_per_launch"] > 0


def test_latest_profile_picks_the_newest_round(tmp_path, monkeypatch):
    (tmp_path / "profiles").mkdir()
    for r in ("r03", "r05", "r10"):
        (tmp_path / "profiles" / f"{r}_pmc_x.json").write_text("{}")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.latest_profile("pmc_x.json") == "r10_pmc_x.json" and bench.latest_profile("pmc_y.json") is None
