"""The host parsers of untrusted bytes (csrc/gk_bamread.cpp, gk_sampack.cpp, gk_textout.cpp) built for the CPU with
AddressSanitizer + UndefinedBehaviorSanitizer and driven with valid and mutated BGZF / BAM / SAM inputs
(tests/asan/driver.py).  Sanitizers run on the CPU build only; the GPU code is not involved."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kir_graph_amd", "csrc")


def _libasan():
    hits = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))
    return hits[-1] if hits else None


@pytest.fixture(scope="module")
def sanitized_library(tmp_path_factory):
    if _libasan() is None:
        pytest.skip("libasan not installed")
    out = str(tmp_path_factory.mktemp("asan") / "libgraphkir_host_asan.so")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", f"-I{ROOT}/include", f"-I{CSRC}",
           f"{CSRC}/gk_bamread.cpp", f"{CSRC}/gk_sampack.cpp", f"{CSRC}/gk_textout.cpp",
           f"{ROOT}/tests/asan/host_stub.cpp", "-o", out, "-lz", "-ldl", "-lpthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_parsers_survive_mutated_inputs_under_asan_ubsan(sanitized_library, seed):
    if seed == 2:   # the record index cut into many segments: guessed chains, meetings, serial stretches
        os.environ["GK_TEST_HOOKS"] = "bam_segments=9"
    env = dict(os.environ, LD_PRELOAD=_libasan(), GK_PACK_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan", "driver.py"), sanitized_library, "250",
                          str(seed)], env=env, capture_output=True, text=True, timeout=900)
    os.environ.pop("GK_TEST_HOOKS", None)
    assert res.returncode == 0 and "OK mutants" in res.stdout, (res.stdout[-500:], res.stderr[-4000:])
