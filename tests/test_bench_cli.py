"""bench.py's launcher / argument / aggregation path on a CPU-only box: `python bench.py --gpus 2` must start two ranks itself,
and a rank count that disagrees with --gpus must fail instead of silently reporting n_gpus = 1.  Uses bench.py's --cpu-stub test hook (gloo, stub solver)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_launches_two_ranks_and_aggregates():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--nb", "1000", "--cpu-stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["config"]["members_total"] == 2000 and rec["config"]["backend"] == "gloo" and rec["scaling"] == "weak"
    assert rec["data"].startswith("cpu-stub")
    # whole-job aggregate: units of both ranks / max time over ranks
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 * 2 - 2 * 300 * 2000) < 1e-3 * 2 * 300 * 2000
    assert rec["config"]["mean_steps_per_member"] == 300


def test_single_rank_stub_line_has_contract_keys():
    r = _run(["--steps", "1", "--warmup", "0", "--nb", "64", "--cpu-stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in rec
    assert rec["n_gpus"] == 1 and rec["dtype"] == "f64" and rec["vs_baseline"] is None and "workload" in rec["config"]


def test_rank_count_must_match_gpus_flag():
    """A launcher that started 1 rank for --gpus 2 (round 1's silent n_gpus = 1) is an error now."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--nb", "64", "--cpu-stub"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_strong_scaling_mode_splits_one_ensemble_over_the_ranks():
    """--config c4 = BASELINE configs[3] as worded: ONE ensemble split over the ranks (strong scaling), one trajectory gather per solve; prints ranks and scaling."""
    r = _run(["--gpus", "2", "--config", "c4", "--members-total", "1001", "--steps", "2", "--warmup", "1", "--cpu-stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec["scaling"] == "strong" and rec["ranks"] == 2 and rec["n_gpus"] == 2
    assert rec["config"]["members_total"] == 1001 and rec["config"]["members_per_gpu"] == 501 and rec["checks"]["finite_and_complete"]
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 * 2 - 2 * 90 * 1001) < 1e-3 * 2 * 90 * 1001


def test_last_stdout_line_is_the_compact_contract_line():
    """the driver keeps an 8 KB stdout tail and parses the LAST line; a 23.7 KB line did not parse.  The last line must be <= 4 KB, carry the contract keys,
    and the full record must sit on an EARLIER line (and in bench_detail.json)."""
    r = _run(["--steps", "1", "--warmup", "0", "--nb", "64", "--cpu-stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines[-1]) <= 4096 and lines[-1].startswith("{")
    rec = json.loads(lines[-1])
    assert rec["detail"] == "bench_detail.json" and "workload" in rec["config"]
    detail = [l for l in lines[:-1] if l.startswith("bench_detail: ")]
    assert len(detail) == 1 and json.loads(detail[0][len("bench_detail: "):])["value"] == json.load(open(os.path.join(ROOT, "bench_detail.json")))["value"]


def test_compact_line_of_a_full_shape_record_fits_4k():
    """the real shape: round 4's full record (every config with roofline + thread-swept CPU leg, 23.7 KB) must compact to <= 4 KB with value, roofline, cpu_baseline
    and one row per config; and a synthetic record with over-long strings everywhere still fits."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line) <= 4096
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "dtype", "scaling", "config", "roofline", "cpu_baseline", "configs"):
        assert k in rec, k
    for k in ("kernel", "bound", "avg_launch_us", "achieved", "peak", "unit", "frac", "lane_ops_frac", "traffic", "counters_from"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "threads", "cores", "usable_cpus", "parallel_efficiency", "seconds", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert set(full["configs"]) <= set(rec["configs"]) and all(len(v) == 4 for v in rec["configs"].values())
    assert abs(rec["value"] / full["value"] - 1) < 1e-4 and abs(rec["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-4
    # hostile shape: long strings, many configs
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 5000
    fat["roofline"]["kernel"] = "k" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    for i in range(12):
        fat["configs"][f"extra_config_number_{i}"] = {"error": "e" * 3000}
    assert len(bench.compact_line(fat)) <= 4096
