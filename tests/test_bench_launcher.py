"""bench.py's rank handling without a GPU: the documented entry point either runs N ranks or fails loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_must_match_the_launchers_world_size():
    """Under a launcher (WORLD_SIZE set) `--gpus` that disagrees is refused before anything touches the GPU."""
    res = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert res.returncode == 2
    assert "does not match WORLD_SIZE" in res.stderr
    assert not res.stdout.strip()


def test_without_a_launcher_the_ranks_are_started_and_a_dead_rank_fails_the_run():
    """`python bench.py --gpus 2` starts two rank processes itself; here they find no GPU and exit, and the
    parent reports the failure instead of printing a line for fewer ranks."""
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "2000", "--cpu-pairs", "0"],
               {})
    assert res.returncode == 1
    assert "rank exit codes" in res.stderr and "fewer than 2 ranks finished" in res.stderr
    assert not res.stdout.strip()


def test_committed_pmc_traffic_belongs_to_the_device_sources_in_the_tree():
    """bench.py reports `roofline.traffic` only from a committed PMC pass whose recorded digest of the device sources
    equals the running code's (bench.measured_traffic): an edit to a kernel's source after the last profile run would
    silently turn the figure into null.  The dominant kernels of the three workloads of the default line must resolve."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    for kernel, pairs, method in (("compat_kernel", 10_000_000, "exonfirst"), ("tab_count", 10_000_000, "em"),
                                  ("compat_kernel", 1_000_000, "pv"), ("setsum_leaves", 10_000_000, "exonfirst"),
                                  ("em_sets_groups", 10_000_000, "em")):
        traffic, why = bench.measured_traffic(kernel, pairs, method)
        assert traffic and traffic > 0, (kernel, pairs, method, why)
