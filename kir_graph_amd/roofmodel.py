"""
Algorithmic bytes / operations of the hot kernels (DESIGN.md section "Roofline accounting").

Every model takes the launch geometries recorded by ``engine`` (``Device.call_log``) and returns
``(bytes, ops)`` per launch -- lower bounds: what the kernel must read and write once, and the
arithmetic it cannot avoid.  ``dominant`` picks the kernel with the largest time per step and prices it
against the roof that bounds it.

``maxsum_chunks`` (one launch = one copy-number step of one gene): reads, once, the candidate columns and
the columns of the previous sets of the gene's log-likelihood table ``L`` (f64, column-major
``[allele][read]``) and writes the ``T x A`` scores:

    bytes = 8 * R * (n_cols + n_prev_cols) + 8 * n_sets * n_cols
    ops   = 2 * R * (outputs computed)         (one f64 max + one f64 add per read and output)

A symmetric launch (the second allele of the search: set t IS column t) computes the tiles on or above the
diagonal only (gk_search.hip), so ``outputs computed`` counts the 32 x 32 tiles with tile_a >= tile_t.
It is a (max,+) contraction: f64 VALU bound, the HBM figure is reported next to it.

``minsum_sad`` (integer bound of the same step): u8 mismatch counts, 4 reads per v_sad_u8.
``compat_kernel``: 4 VALU lane-operations per (id, allele) -- VALU bound (65 % VALU-busy measured), HBM beside it.
VALU peaks are priced per instruction at the issue cost measured on the device (``ISSUE_CYCLES``), with the guide's
figure beside them (``GUIDE_CYCLES``).
``tab_count``: a byte stream, HBM bound by construction (divergent walk: far below the roof).
"""
from __future__ import annotations

HBM_PEAK_GBS = 8000.0
TILE = 32
N_SIMD = 256 * 4
NOMINAL_HZ = 2.4e9

# Issue cost of the VALU instructions the hot kernels are made of, MEASURED on MI355X with tools/valu_rate.hip at 8 waves
# per SIMD (profiles/r03_valu_rate.txt): cycles of a 2.4 GHz clock per wave64 instruction and SIMD.  (The shader clock
# under that load read 2.21 - 2.23 GHz, so 4.5 of these "cycles" are ~4.1 real ones: every VOP3-encoded 32-bit
# instruction measured costs what a float64 instruction costs; only plain VOP2 forms -- v_mul_f32 2.40, v_and_b32 2.83,
# v_add_u32 3.02 -- come near the 2 cycles of the guide.)
ISSUE_CYCLES = {"v_mul_f64": 4.50, "v_add_f64": 4.26, "v_max_f64": 4.19, "v_bfe_i32": 4.43, "v_bfi_b32": 4.56,
                "v_sad_u8": 4.49}
# MI355X_MICROARCH.md: "a wave issues each VALU instruction over 2 cycles (32 lanes/cycle x 2)" with >= 2 waves per
# SIMD (157.3 TFLOP/s fp32 = 78.6 T lane-FMAs/s); float64 at half that rate (78.6 TFLOP/s = 39.3 T lane-FMAs/s)
GUIDE_CYCLES = {"v_mul_f64": 4.0, "v_add_f64": 4.0, "v_max_f64": 4.0, "v_bfe_i32": 2.0, "v_bfi_b32": 2.0, "v_sad_u8": 2.0}
# The same instructions issued as the SEQUENCE the kernel runs (tools/valu_rate.hip "compat factor": v_bfe_i32 + 2 x v_bfi_b32
# + v_mul_f64 on eight independent chains per wave, 8 waves per SIMD): 5.81 cycles per instruction -- any mix of these VOP3
# and float64 instructions issues there, whatever the dependencies -- where the four cost 4.51 on average one kind at a
# time.  Reported beside `peak` as `peak_sequence` (profiles/r03_valu_rate.txt).
SEQUENCE_CYCLES = {"compat_kernel": 5.81}
# the instructions one algorithmic operation of a kernel cannot do without
KERNEL_OPS = {
    "compat_kernel": ("v_bfe_i32", "v_bfi_b32", "v_bfi_b32", "v_mul_f64"),   # per (id, allele): bit -> mask, two halves, product
    "minsum_sad": ("v_sad_u8",),                                              # per 4 reads, set and allele
    "maxsum_chunks": ("v_max_f64", "v_add_f64"),                              # per read, set and allele
}


def laneOpsPeak(kernel: str, table: dict[str, float] = ISSUE_CYCLES) -> float:
    """Lane-operations per second of the whole GPU when the kernel's instruction mix issues back to back at the costs of
    ``table``: (instructions x 64 lanes x 1024 SIMDs) / (sum of their issue cycles / 2.4 GHz)."""
    ops = KERNEL_OPS[kernel]
    return len(ops) * 64 * N_SIMD * NOMINAL_HZ / sum(table[o] for o in ops)


def symmetricOutputs(n: int) -> int:
    """Outputs a symmetric maxsum launch computes: (t, a) with tile(a) >= tile(t), both < n."""
    total = 0
    for t0 in range(0, n, TILE):
        rows = min(TILE, n - t0)
        total += rows * (n - t0)          # columns of tiles t0 .. end
    return total


def maxsumLaunch(n_rows: int, n_sets: int, c_prev: int, n_cols: int, n_prev_cols: int,
                 symmetric: bool = False) -> tuple[float, float]:
    by = 8.0 * n_rows * (n_cols + n_prev_cols) + 8.0 * n_sets * n_cols
    outputs = symmetricOutputs(n_cols) if symmetric else n_sets * n_cols
    return by, 2.0 * n_rows * outputs


def minsumLaunch(n_rows: int, n_sets: int, n_cols: int, n_prev_cols: int, symmetric: bool = False
                 ) -> tuple[float, float]:
    """u8 tables: bytes = R * (n_cols + n_prev_cols) + 4 * outputs; ops = one v_sad_u8 lane-op per 4 reads
    and output."""
    outputs = symmetricOutputs(n_cols) if symmetric else n_sets * n_cols
    return 1.0 * n_rows * (n_cols + n_prev_cols) + 4.0 * n_sets * n_cols, n_rows * outputs / 4.0


def compatLaunch(n_rows: int, n_allele: int, n_ids: float, out_bytes: int = 8) -> tuple[float, float]:
    """reads the rows' id lists (4 B per id, 16 B of offsets per row), writes the table (8 B per entry as float64, 2 B in
    the index form) and the mismatch byte; per id and allele the floor of the per-lane formulation is 4 VALU
    lane-operations (bit -> mask, two half-word selects of 0.999 / 0.001, one f64 multiply: the ordered product cannot
    be reassociated)."""
    return 4.0 * n_ids + 16.0 * n_rows + float(out_bytes + 1) * n_rows * n_allele, 4.0 * n_ids * n_allele


def tabLaunch(n_pairs: int, n_valid: int, n_ids: int) -> tuple[float, float]:
    """reads 2 x 128-byte records per pair, writes the four id lists."""
    return 256.0 * n_pairs + 4.0 * n_ids + 22.0 * n_valid, 0.0


def setsumLaunch(n_rows: int, n_sets: int, c: int, n_distinct: int) -> tuple[float, float]:
    """exact value + shares of the contender sets: reads their distinct columns once; c max + 1 add per read, set."""
    return 8.0 * n_rows * n_distinct + 8.0 * n_sets * (c + 1), float(n_rows) * n_sets * (c + 1)


def emSetsLaunch(n_rows: int, n_ids: float, set_words: int) -> tuple[float, float]:
    """candidate sets of the pairs of all genes of a sample (one launch): per pair its position (4 B), five list offsets
    (20 B) and its ids (4 B each) read, its set (4 B per word) written; the bit rows come out of LDS."""
    return 24.0 * n_rows + 4.0 * n_ids + 4.0 * set_words, 0.0


def emVerifyLaunch(set_words: int) -> tuple[float, float]:
    """every set read once more and compared with the first set of its hash (those come out of L2)."""
    return 4.0 * set_words, 0.0


def _priced(kernel: str, calls: list[tuple]) -> tuple[float, float, str, float]:
    """(bytes, ops, bound, ops peak) summed over the recorded launches of ``kernel``."""
    by = ops = 0.0
    bound, peak = "hbm", 0.0
    for c in calls:
        if kernel == "maxsum_chunks":
            b, o = maxsumLaunch(*c[1:7])
            bound, peak = "valu", laneOpsPeak("maxsum_chunks")
        elif kernel == "colsum_chunks":      # column sums: one add per element, a pure HBM stream
            b, o = 8.0 * c[1] * c[4] + 8.0 * c[4], 0.0
        elif kernel == "em_sets_groups":
            b, o = emSetsLaunch(*c[1:4])
        elif kernel == "em_sets_verify":
            b, o = emVerifyLaunch(c[3])
        elif kernel == "minsum_sad":
            b, o = minsumLaunch(*c[1:6])
            bound, peak = "valu", laneOpsPeak("minsum_sad")
        elif kernel == "compat_kernel":
            b, o = compatLaunch(*c[1:5])
            bound, peak = "valu", laneOpsPeak("compat_kernel")
        elif kernel == "tab_count":
            b, o = tabLaunch(*c[1:4])
        elif kernel in ("fraction_chunks", "setsum_leaves"):
            b, o = setsumLaunch(*c[1:5])
        else:
            return 0.0, 0.0, "hbm", 0.0
        by += b
        ops += o
    return by, ops, bound, peak


def summarise(call_log: list[tuple], kernel: str, total_ms: float, launches: int) -> dict:
    """Roofline entry for ``kernel`` from the recorded launch geometries and its HIP-event time."""
    calls = [c for c in call_log if c[0] == kernel]
    base = {"kernel": kernel, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None, "traffic": None, "launches": launches, "avg_launch_ms": total_ms / max(launches, 1)}
    if not calls or total_ms <= 0:
        return base
    by, ops, bound, peak = _priced(kernel, calls)
    if by <= 0:
        return base
    sec = total_ms / 1e3
    # the log holds one entry per API call; a call may issue more than one launch of the kernel (e.g. the
    # two template variants of maxsum), so per-launch figures divide by the launches the events counted
    gbs = by / sec / 1e9
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    out = dict(base)
    out["algorithmic_bytes_per_launch"] = by / max(launches, 1)
    if bound == "valu":
        tops = ops / sec / 1e12
        unit = {"maxsum_chunks": "Tops/s f64 (max+add)", "minsum_sad": "T lane-ops/s (v_sad_u8, 4 reads each)",
                "compat_kernel": "T lane-ops/s (4 VALU per id and allele)"}.get(kernel, "T lane-ops/s")
        guide = laneOpsPeak(kernel, GUIDE_CYCLES)
        out.update({"bound": "valu", "achieved": tops, "peak": peak / 1e12, "unit": unit, "frac": ops / sec / peak,
                    "peak_guide": guide / 1e12, "frac_guide": ops / sec / guide,
                    "issue_cycles": {o: ISSUE_CYCLES[o] for o in KERNEL_OPS[kernel]},
                    "algorithmic_ops_per_launch": ops / max(launches, 1), "hbm": hbm,
                    "note": "VALU-issue bound (ordered f64 products / a (max,+) or (min,+) contraction).  `peak` = the "
                            "kernel's own instruction mix issued back to back at the MEASURED cost of each instruction "
                            "(tools/valu_rate.hip at 8 waves per SIMD, profiles/r03_valu_rate.txt: cycles of a 2.4 GHz "
                            "clock per wave64 instruction and SIMD in `issue_cycles`); `peak_guide` / `frac_guide` price "
                            "the same mix at the guide's 2 cycles per 32-bit and 4 per float64 instruction.  The HBM "
                            "roof is shown beside it; symmetric launches are credited with the triangle of tiles they "
                            "compute"})
        if kernel in SEQUENCE_CYCLES:
            seq = 64 * N_SIMD * NOMINAL_HZ / SEQUENCE_CYCLES[kernel]
            out["peak_sequence"] = seq / 1e12
            out["frac_sequence"] = ops / sec / seq
            out["note"] += (".  `peak_sequence` / `frac_sequence`: the same instructions at the rate measured for them "
                            "as the sequence the kernel issues (eight independent chains per wave, 8 waves per SIMD)")
    else:
        out.update(hbm)
        out["bound"] = "hbm"
    return out


def dominant(prof: dict[str, tuple[int, float]], call_log: list[tuple]) -> dict:
    """Roofline entry of the kernel with the largest total time in ``prof`` ({name: (launches, ms)})."""
    if not prof:
        return summarise([], "none", 0.0, 0)
    # the kernel with the largest time among those this module prices (tiny samples put an unpriced launch-bound kernel
    # on top: it is named beside the entry)
    priced = {k: v for k, v in prof.items() if _priced(k, [c for c in call_log if c[0] == k])[0] > 0}
    top = max(prof.items(), key=lambda kv: kv[1][1])[0]
    name, (launches, total_ms) = max((priced or prof).items(), key=lambda kv: kv[1][1])
    out = summarise(call_log, name, total_ms, launches)
    total = sum(v[1] for v in prof.values())
    out["share_of_kernel_time"] = total_ms / total if total else None
    if top != name:
        out["largest_kernel_by_time"] = top
    # the other priced kernels, for the record
    others = {}
    for k, (n, ms) in prof.items():
        if k != name and ms > 0.02 * total:
            e = summarise(call_log, k, ms, n)
            if e.get("achieved") is not None:
                others[k] = {f: e[f] for f in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
    if others:
        out["other_kernels"] = others
    return out


def stepRoofline(call_log: list[tuple], steps: int, ms_per_step: float) -> dict:
    """The whole step against the roofs: the algorithmic bytes of EVERY priced launch of a step (tabulation, compatibility
    tables, bounds, exact sums, column sums) over the step's measured wall time against the HBM peak -- the metric's
    "achieved HBM GB/s vs roofline" as one number -- and, beside it, how much of the step the VALU-bound kernels' own
    floors take (algorithmic lane-operations at the measured issue peak of each kernel's instruction mix).
    ``call_log`` covers ``steps`` steps (the serial pass); ``ms_per_step`` is the reported one (pipelined legs)."""
    kernels = sorted({c[0] for c in call_log})
    total_bytes = hbm_floor = valu_floor = valu_floor_guide = 0.0
    per_kernel = {}
    for k in kernels:
        calls = [c for c in call_log if c[0] == k]
        by, ops, bound, peak = _priced(k, calls)
        if by <= 0:
            continue
        total_bytes += by
        hbm_floor += by / (HBM_PEAK_GBS * 1e9)
        if bound == "valu" and peak > 0:
            valu_floor += ops / peak
            valu_floor_guide += ops / laneOpsPeak(k, GUIDE_CYCLES)
        per_kernel[k] = {"bytes_per_step": by / max(steps, 1), "ops_per_step": ops / max(steps, 1), "bound": bound}
    steps = max(steps, 1)
    sec = ms_per_step / 1e3
    gbs = total_bytes / steps / sec / 1e9 if sec > 0 else None
    return {"algorithmic_bytes_per_step": total_bytes / steps, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS if gbs is not None else None,
            "hbm_floor_ms_per_step": 1e3 * hbm_floor / steps,
            "valu_floor_ms_per_step": 1e3 * valu_floor / steps,
            "valu_share_of_step": (1e3 * valu_floor / steps) / ms_per_step if ms_per_step > 0 else None,
            "valu_floor_ms_per_step_guide": 1e3 * valu_floor_guide / steps,
            "kernels": per_kernel,
            "note": "sum over all priced launches of one step (serial pass geometries) / the reported ms_per_step; the "
                    "step is VALU-issue bound, not HBM bound: valu_floor = the VALU-bound kernels' algorithmic "
                    "lane-operations at the measured issue peak of their instruction mix (…_guide: at the guide's price)"}
