"""bench.py run the way the driver runs it (a child process, small sample): the one JSON line and its contract -- the
metric of BASELINE.json, whole-job throughput from records in pinned host memory, the legs from records
resident in HBM beside it, the median of the timed legs, the roofline of the dominant kernel measured in the same run, the
host budget, what the value table did -- and the multi-rank form on one GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # ONE JSON line on stdout
    return json.loads(lines[0])


def test_one_gpu_line_keeps_the_contract(device):
    d = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--pairs", "20000", "--cpu-pairs", "0", "--serial-steps", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "host", "hbm_resident", "legs",
                "value_table_new_per_sample", "samples_repeated_pass", "search_steps"):
        assert key in d, key
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["inputs"] == "host" and d["config"]["pairs_per_sample"] == 20000
    assert abs(d["value"] - 2 * 20000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9      # whole-job reads per second
    # three timed legs, the median one is the line's
    assert len(d["legs"]) == 3 and sorted(x["ms_per_step"] for x in d["legs"])[1] == pytest.approx(d["ms_per_step"])
    # 8 distinct samples, 2 of them warmed up: the first leg meets 4 new ones, the later legs none
    assert d["distinct_samples"] == 8 and d["legs"][0]["value_table_new"] >= 0 and d["legs"][2]["value_table_new"] == 0
    assert d["legs"][2]["samples_repeated_pass"] == 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["unit"] and r["peak"] > 0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and "kernel" in r and r["avg_launch_ms"] > 0
    h = d["host"]
    assert h["host_core_s_per_step"] > 0 and h["worker_processes"] == 1
    # lanes x searches follow the host cores the rank has (cohort.pipelineDefaults)
    assert (h["sample_lanes"], h["search_slots"]) in ((5, 3), (4, 2), (3, 2))
    p = d["hbm_resident"]
    assert p["unit"] == "reads/s" and p["value"] > 0 and p["ms_per_step"] > 0 and len(p["legs"]) == 3
    assert d["search_steps"]["bounded"] > 0


def test_steps_from_hbm_only(device):
    d = _run(["--steps", "4", "--warmup", "2", "--pairs", "20000", "--cpu-pairs", "0", "--serial-steps", "0", "--inputs", "hbm",
              "--one-kind", "--legs", "1"])
    assert d["config"]["inputs"] == "hbm" and "pcie_inclusive" not in d and "hbm_resident" not in d and len(d["legs"]) == 1
    assert d["host"]["worker_processes"] == 1 and d["value"] > 0


def test_default_run_reports_the_three_workloads(device):
    """Without --pairs / --method the line is configs[2] exon-first with `em` and `configs1_pv` beside it, each with its
    own legs, serial kernel table, roofline and CPU baseline (here at 1 / 500 of the sizes: the structure, not the
    numbers; the names of the kernels are their own, so the EM kernels and the set sums show under them)."""
    d = _run(["--steps", "3", "--warmup", "1", "--cpu-pairs", "5000", "--serial-steps", "1", "--legs", "1", "--cli-samples", "0",
              "--pairs-scale", "0.002"], timeout=900)
    assert d["config"]["allele_strategy"] == "exonfirst" and d["config"]["pairs_per_sample"] == 20000
    assert d["n_gpus"] == 1 and d["value"] > 0 and "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port"
    for name, method, pairs in (("em", "em", 20000), ("configs1_pv", "pv", 2000)):
        o = d[name]
        assert o["workload"] == name and o["config"]["allele_strategy"] == method and o["config"]["pairs_per_sample"] == pairs
        assert o["value"] > 0 and o["unit"] == "reads/s" and len(o["legs"]) == 1
        assert o["kernels_serial"]["kernels"] and o["roofline"]["kernel"] in o["kernels_serial"]["kernels"]
        assert o["cpu_baseline"]["value"] > 0 and o["cpu_baseline"]["cores"] == 1
    em_kernels = d["em"]["kernels_serial"]["kernels"]
    assert {"em_sets_groups", "em_sets_verify", "em_sets_emit", "em_kernel_genes"} <= set(em_kernels)
    assert "compat_kernel" in d["kernels_serial"]["kernels"] and "compat_kernel" not in em_kernels
    # kernels are timed under their own names: no collective labels left
    assert not {"select_cut", "select", "scan", "count_ids"} & set(d["kernels_serial"]["kernels"])


def test_two_ranks_on_one_gpu_through_the_file_backend(device):
    d = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--pairs", "20000", "--cpu-pairs", "0", "--serial-steps", "0",
              "--one-kind"], env={"GK_COMM_BACKEND": "file"})
    assert d["n_gpus"] == 2 and d["config"]["rank_barrier"] == "file" and "hbm_resident" not in d
    assert abs(d["value"] - 2 * 2 * 20000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9       # both ranks' reads / the slower rank's time


def test_cli_typing_stage_keeps_pace_with_the_bench(device):
    """The command line's typing stage (main.alleleTyping: copy-number files, typing lanes, the two files per sample) on 12
    configs[1] samples against the bench's step on the same samples in the same process: one code path
    (cohort.SampleTyper), so the CLI may cost at most 1.3 x the measured step."""
    # three lanes for both: with five (the default where a rank has six cores or more) the fill and drain of the pipeline
    # are a third of a 12-sample run, which says nothing about the code path
    d = _run(["--steps", "12", "--warmup", "4", "--pairs", "1000000", "--method", "pv", "--distinct", "4", "--cpu-pairs", "0",
              "--serial-steps", "0", "--legs", "1", "--cli-samples", "12"], env={"GK_SAMPLE_LANES": "3", "GK_SEARCH_SLOTS": "2"}, timeout=900)
    c = d["cli_typing_stage"]
    assert c["samples"] == 12 and c["ms_per_sample"] > 0
    assert c["ms_per_sample"] <= 1.3 * d["ms_per_step"], (c["ms_per_sample"], d["ms_per_step"])
