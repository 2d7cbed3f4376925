"""The library's staging rings (csrc/gk_runtime.hip: gk_send / gk_fetch_queue): transfers below 4 MB go through two
pinned rings of 8 MB whose space comes back when a mark of the stream has passed or the stream has been drained; the
whole-sample search keeps dozens of them in flight (gk_sample_search).  Here: enough traffic to go round both rings
several times, sizes that do not divide them, and the bytes that come back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_staging_rings_wrap_without_losing_bytes(device):
    rng = np.random.default_rng(7)
    sizes = [1, 63, 64, 65, 4093, 1 << 16, (1 << 20) + 17, (3 << 20) + 5, (4 << 20) - 64, (4 << 20), (4 << 20) + 8]
    kept = []
    for rep in range(6):                       # ~ 110 MB each way: both 8 MB rings wrap a dozen times
        for n in sizes:
            a = rng.integers(0, 256, size=n, dtype=np.uint8)
            kept.append((a, device.put(a)))
    for a, buf in kept:
        assert np.array_equal(buf.download(), a)
    # many small transfers in a row (parameters of the search steps are a few KB each)
    small = [rng.integers(0, 1 << 30, size=rng.integers(1, 3000), dtype=np.int32) for _ in range(4000)]
    bufs = [device.put(a) for a in small]
    for a, b in zip(small, bufs):
        assert np.array_equal(b.download(), a)
    for _, b in kept:
        b.free()
    for b in bufs:
        b.free()
