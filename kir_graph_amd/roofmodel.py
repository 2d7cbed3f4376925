"""
Algorithmic bytes / operations of the hot kernels (DESIGN.md section "Roofline accounting").

``maxsum_chunks`` (one launch = one copy-number step of one gene): it must read, once, the
candidate columns and the columns of the previous sets of the gene's log-likelihood table
``L`` (f64, column-major ``[allele][read]``) and write the ``T x A`` scores:

    bytes = 8 * R * (n_cols + n_prev_cols) + 8 * n_sets * n_cols
    ops   = 2 * R * n_sets * n_cols            (one f64 max + one f64 add per read, set, column)

These are lower bounds (no re-reads counted); the kernel re-reads L tiles through L2/MALL.
"""
from __future__ import annotations

F64_VALU_PEAK_OPS = 78.6e12 / 2   # v_max_f64 / v_add_f64 issue rate = half the FMA-counted 78.6 TFLOP/s
HBM_PEAK_GBS = 8000.0


def maxsumLaunch(n_rows: int, n_sets: int, c_prev: int, n_cols: int, n_prev_cols: int) -> tuple[float, float]:
    by = 8.0 * n_rows * (n_cols + n_prev_cols) + 8.0 * n_sets * n_cols
    ops = 2.0 * n_rows * n_sets * n_cols
    return by, ops


def summarise(call_log: list[tuple], kernel: str, total_ms: float, launches: int) -> dict:
    """Roofline entry for ``kernel`` from the recorded launch geometries and its HIP-event time."""
    calls = [c for c in call_log if c[0] == kernel]
    if kernel != "maxsum_chunks" or not calls or total_ms <= 0:
        return {"kernel": kernel, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None, "traffic": None, "launches": launches,
                "avg_launch_ms": total_ms / max(launches, 1)}
    by = ops = 0.0
    for _, r, t, c, a, pc in calls:
        b, o = maxsumLaunch(r, t, c, a, pc)
        by += b
        ops += o
    sec = total_ms / 1e3
    gbs = by / sec / 1e9
    return {
        "kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": gbs / HBM_PEAK_GBS, "traffic": None, "launches": launches,
        "avg_launch_ms": total_ms / max(launches, 1), "algorithmic_bytes_per_launch": by / len(calls),
        "note": "the (max,+) contraction is f64-VALU bound, not HBM bound; see valu",
        "valu": {"achieved": ops / sec / 1e12, "peak": F64_VALU_PEAK_OPS / 1e12, "unit": "Tops/s f64 (max+add)",
                 "frac": ops / sec / F64_VALU_PEAK_OPS},
    }
