"""CPU: the bench line committed under profiles/ (written by bench.py on an MI355X, scripts/collect_profiles.sh)
carries every field the measurement contract names, with consistent arithmetic."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip(name + " not collected")
    return json.loads(open(path).read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r02_bench_n1.json", "r03_bench_n1.json"])
def test_default_bench_line_contract(name):
    j = _line(name)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"].split(",")[0] == base["metric"].split(",")[0]          # BASELINE's metric
    assert j["unit"] == "elements/s" and j["higher_is_better"] is True and j["n_gpus"] == 1
    assert j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    # value = elements of the K steps / their time
    ne = j["config"]["elements_total"]
    assert abs(j["value"] - ne / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_us_avg"):
        assert key in r, key
    assert r["unit"] == "TFLOP/s" and r["bound"] in ("fp64-valu", "mfma", "hbm")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    # achieved = algorithmic flops per launch / the kernel's average launch duration
    flops = r["flops_per_element"] * r["elements_per_launch"]
    assert abs(r["achieved"] - flops / (r["kernel_us_avg"] * 1e-6) / 1e12) <= 1e-9 * r["achieved"]
    # the dominant kernel fits inside the step it dominates -- up to the part of a dispatch's begin -> end
    # stamps that overlaps the previous launch in a back-to-back sequence: an EMPTY kernel reads 4.1 us
    # stamped and 3.1 us per launch back to back (profiles/r03_launch_floor.txt), so K back-to-back steps of a
    # ~5 us kernel take up to ~1 us per step LESS than K stamped durations (roofline.kernel_us_is says so)
    assert r["kernel_us_avg"] * 1e-3 <= max(1.05 * j["ms_per_step"], j["ms_per_step"] + 1.2e-3)
    # how the K steps were timed is part of the line: warm steady state first, the replay of the captured steps
    # timed, the loop issued call by call measured beside it on the same buffers
    if name.startswith("r03"):                  # (fields of the round-3 line)
        assert j["prewarm_steps"] >= 9
        assert j["config"]["timed_region"].startswith("the K steps captured once in a hipGraph")
        e = j["eager_loop"]
        assert e["results_equal"] is True and e["mode_used"] == "eager"
        assert 0.7 * j["ms_per_step"] <= e["ms_per_step"] <= 1.5 * j["ms_per_step"]
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert j["value"] >= 1.0e6                                              # north_star's target
    if name.startswith("r03"):
        # round 3: the nominal figure is labelled, the EXECUTED issue-slot fraction stands beside it
        assert "direct-Gram-EQUIVALENT" in r["achieved_is"]
        ex = r["executed"]
        assert 0.0 < ex["issue_slot_frac"] <= 1.0 and ex["issue_slot_frac"] <= r["frac"]
        assert r["bound"] == "fp64-valu"


@pytest.mark.parametrize("name", ["r02_bench_deg32.json", "r02_bench_1e7.json", "r02_bench_dual_deg8.json",
                                  "r03_bench_deg32.json", "r03_bench_1e7.json", "r03_bench_dual_deg8.json",
                                  "r03_bench_dual_deg32.json"])
def test_other_bench_lines_are_consistent(name):
    j = _line(name)
    r = j["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and 0.0 < r["frac"] < 1.5
    assert j["n_gpus"] == 1 and j["steps"] >= 1 and j["unit"] == "elements/s"
    if name.startswith("r03"):
        assert 0.0 < r["executed"]["issue_slot_frac"] <= 1.0        # what the kernel executes can never exceed the pipe
        if "dual" in name:
            assert r["bound"] == "fp64-valu" and "MFMA" not in r["pipe"].replace("no MFMA", "")


def test_config5_bench_line_contract():
    """BASELINE config 5 (`bench.py --config 5`): an HBM-bound line -- achieved = algorithmic bytes per launch /
    the kernel's average launch duration, both roofs reported, accuracy against the 60-digit minimiser, the SLSQP
    baseline with the variable-coefficient residual, PMC traffic from profiles/traffic.json."""
    j = _line("r03_bench_c5.json")
    assert j["unit"] == "elements/s" and j["n_gpus"] == 1 and j["dtype"] == "f64" and j["vs_baseline"] is None
    assert "variable-coefficient" in j["metric"] and "workload" in j["config"]
    ne = j["config"]["elements_total"]
    assert ne in (1000008, 1000000)
    assert abs(j["value"] - ne / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["bytes_per_element"] == 472
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - 472 * r["elements_per_launch"] / (r["kernel_us_avg"] * 1e-6) / 1e9) <= 1e-9 * r["achieved"]
    assert r["kernel_us_avg"] * 1e-3 <= 1.05 * j["ms_per_step"]
    f = j["roofline_fp64"]
    assert abs(f["frac"] - f["achieved"] / f["peak"]) <= 1e-12 and 0.0 < f["frac"] <= 1.0
    assert 0.0 < f["executed"]["issue_slot_frac"] <= 1.0
    assert j["accuracy"]["rel_l2_vs_60_digit_minimiser"] <= 1e-13
    assert j["accuracy"]["rel_l2_bubble_vs_60_digit_minimiser"] <= 1e-13
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "Dual.py:43-44" in c["sample"]
    assert j["other_table_layout"]["W_bit_equal_to_primary_layout"] is True
    assert j["value"] >= 1.0e6
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["c5_M9_n16_ne1000008"]
    assert 0.9 <= t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"] <= 1.25


def test_round4_default_line_contract():
    """Round 4: `roofline` prices the kernel the TIMED REGION launches with the region's own per-step time (the
    round-3 line priced the enhancement-only kernel stamped elsewhere: 0.46 against 0.41), `value` is the median of
    >= 10 brackets of exactly K steps, and BASELINE configs 4 and 5 ride along in the default line with their own
    roofline and CPU baseline."""
    j = _line("r04_bench_n1.json")
    ne = j["config"]["elements_total"]
    assert ne == 100008 and abs(j["value"] - ne / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert j["timed_brackets"] >= 10 and j["ms_per_step_min"] <= j["ms_per_step"] <= j["ms_per_step_max"]
    # the MEDIAN is what `value` uses: within 5 % of the best bracket; a single slow bracket (a host hiccup inside an
    # eager K-step loop: 10.4 against 8.0 us in the line of record) is what the median is there to absorb
    assert j["ms_per_step"] <= 1.05 * j["ms_per_step_min"] and j["ms_per_step_max"] <= 1.5 * j["ms_per_step_min"]
    r = j["roofline"]
    assert r["kernel"].startswith("step_small_kernel<M=9>") and r["bound"] == "fp64-valu"
    assert abs(r["kernel_us_avg"] - j["ms_per_step"] * 1e3) <= 1e-9 * r["kernel_us_avg"]
    flops = r["flops_per_element"] * r["elements_per_launch"]
    assert abs(r["achieved"] - flops / (r["kernel_us_avg"] * 1e-6) / 1e12) <= 1e-9 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and r["peak"] == 78.6
    eo = r["enhancement_only"]
    assert eo["kernel"].startswith("enhance_small_kernel") and eo["kernel_us_in_sequence_avg"] <= 1.05 * r["kernel_us_avg"]
    ex = r["executed"]
    assert 0.0 < ex["issue_slot_frac"] <= 1.0 and "r04" in ex["source"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    for key, bound, unit in (("config4", "fp64-valu", "TFLOP/s"), ("config5", "hbm", "GB/s")):
        c = j[key]
        cr = c["roofline"]
        assert cr["bound"] == bound and cr["unit"] == unit and abs(cr["frac"] - cr["achieved"] / cr["peak"]) <= 1e-12
        assert abs(cr["kernel_us_avg"] - c["ms_per_step"] * 1e3) <= 1e-9 * cr["kernel_us_avg"]
        assert abs(c["value"] - c["config"]["elements_total"] / (c["ms_per_step"] * 1e-3)) <= 1e-6 * c["value"]
        assert c["timed_brackets"] >= 10 and c["value"] >= 1.0e6
        cb = c["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["wall_s"] < 60.0
        assert c["accuracy"]["rel_l2_vs_60_digit_minimiser"] <= 1e-13
        assert c["accuracy"]["rel_l2_bubble_vs_60_digit_minimiser"] <= 1e-13
    assert "degree 32 / 64 points" in j["config4"]["cpu_baseline"]["sample"]
    assert "Dual.py:43-44" in j["config5"]["cpu_baseline"]["sample"]
    assert j["config5"]["roofline"]["bytes_per_element"] == 472


def test_round4_multi_gpu_line_carries_roofline_and_cpu_baseline():
    """The N > 1 line (rehearsed with two gloo ranks on one GPU: its rates mean nothing for xGMI) has the keys the
    N = 1 line is judged on: aggregate roofline against N x the one-GPU peaks, HBM beside it, the CPU baseline."""
    j = _line("r04_bench_gloo2_rehearsal.json")
    n = j["n_gpus"]
    assert n == 2 and j["timed_brackets"] >= 10
    r = j["roofline"]
    assert r["peak"] == n * 78.6 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    assert abs(r["kernel_us_avg"] - j["ms_per_step"] * 1e3) <= 1e-9 * r["kernel_us_avg"]
    tot = j["config"]["elements_total"]
    assert abs(r["achieved"] - r["flops_per_element"] * tot / (r["kernel_us_avg"] * 1e-6) / 1e12) <= 1e-9 * r["achieved"]
    assert j["roofline_hbm"]["peak"] == n * 8000.0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
