"""
``graphkir`` command line -- drop-in for the typing stages of ``graphkir/main.py``.

Same flags, defaults, file naming and TSV schemas (createParser 258-420, main 423-606,
readMapping 124-168, alleleTyping 171-220, getCommonName 223-250).  What differs:

* tabulation, read depth, copy-number fit and allele typing run on the GPU; a sample's tabulation
  is handed to typing in memory (the ``.variant.json`` is still written unless
  ``--no-variant-json``);
* additive flags: ``--alignment`` (use existing name-collated SAM/BAM instead of running hisat2),
  ``--no-variant-json``; ``--allele-strategy`` also accepts ``pv`` (= full) and maps ``report`` to
  the EM strategy (the reference forwards ``report`` to a factory that rejects it, main.py:192);
* launched with RANK / WORLD_SIZE / LOCAL_RANK set (``torchrun`` or any launcher) the samples are sharded over
  the ranks (one GPU each, largest inputs first); ``--cn-cohort`` then pools the gene depths with one
  all-gather (``cohort.Comm`` -> ``gk_allgather_f64``, RCCL), rank 0 merges the outputs;
* ``--ranks N`` starts the N rank processes itself (no launcher needed);
* a sample is typed and released as soon as its copy numbers are known (memory does not grow with the cohort);
* index building, WGS extraction and plotting are outside this build: the index files must exist.
"""
from __future__ import annotations

import argparse
import os
import logging
from pathlib import Path

import numpy as np
import pandas as pd

from . import cohort
from .external_tools import setEngine
from .hisat2 import (ParkedRecords, SampleData, extractVariant, extractVariantFromPacked, extractVariantFromText,  # noqa: F401
                     packAlignments, readExons, readPair,
                     saveReadsToBam, writeCompact, writeCompactRecords,
                     writeReadsAndVariantsData, writeSampleJson)
from .index import GkIndex
from .kir_cn import filterDepth, loadCN, predictSamplesCN
from .kir_typing import defaultDevice, selectKirTypingModel
from .samtools_utils import depthOfSample, readLocusLengths
from .utils import testHook, getThreads, logger, mergeAllele, mergeCN, setThreads


def getCommonName(r1: str, r2: str) -> str:
    """Longest common dot-separated prefix of the two read file names (main.py:223-250)."""
    name = ""
    for s1, s2 in zip(r1.split("."), r2.split(".")):
        if s1 != s2:
            return name
        name = name + "." + s1 if name else s1
    return name


def replaceParentFolder(filename: str, new_folder: str) -> str:
    return str(Path(new_folder) / Path(filename).name)


def hisatMap(index: str, f1: str, f2: str, output_file: str, threads: int = 1) -> None:
    """External aligner, unchanged from the reference (hisat2.py:68-92); needs hisat2 + samtools."""
    from .external_tools import runTool
    assert output_file.endswith(".bam")
    name = output_file.rsplit(".", 1)[0]
    runTool("hisat", ["hisat2", "--threads", str(threads), "-x", index, "-1", f1, "-2", f2,
                      "--no-spliced-alignment", "--max-altstried", "64", "--haplotype", "-S", f"{name}.sam"])
    runTool("samtools", ["samtools", "sort", f"-@{threads}", f"{name}.sam", "-o", f"{name}.bam"])
    runTool("samtools", ["samtools", "index", f"-@{threads}", f"{name}.bam"])


def mapSamples(names, reads, index, index_ref, exon_region_only=False, alignments=None, write_json=True,
               keep_records=False):
    """Graph mapping (external) -> tabulation (GPU) -> depth (GPU), one sample at a time (main.py:124-168).

    Yields ``(alignment file, "{name}.variant", SampleData, depth file)`` per sample, in order.  When a sample
    is yielded its by-products are on disk and its depth is computed, so what only they needed -- the SAM
    text of the pairs and the packed records in HBM -- has been released: the consumer holds the CSR of the
    tabulation only (and closes it after typing), whatever the size of the cohort.  ``keep_records``: the packed
    records stay in HBM with the sample (``--cn-cohort`` parks a sample as its compact records, ``ParkedRecords``)."""
    gk = GkIndex.load(index_ref)
    gene_len = readLocusLengths(index_ref)
    dev = defaultDevice()
    from .engine import DeviceIndex
    dindex = DeviceIndex(dev, gk)
    # the staging contexts of the sample pipeline (cohort.stagingContexts, the ones bench.py stages with): the records go
    # to HBM on the copier's stream, the tabulation and the depth run on a high-priority stream of their own -- beside the
    # typing lanes (cohort.SampleTyper), which are typing the samples before this one
    copier, ingest = cohort.stagingContexts(dev)

    def prepare(k: int):
        """External mapping (when needed) + native packing of sample k: host work, off the GPU's path."""
        name = names[k] + "." + index.replace(".", "_").replace("/", "_")
        if alignments:
            source = alignments[k]
        else:
            logger.info(f"[Graph] Run graph mapping on index {index} ({name})")
            hisatMap(index, reads[k][0], reads[k][1], name + ".bam", threads=getThreads())
            source = name + ".bam"
        pack = None
        if source.endswith((".sam", ".sam.gz")) or testHook("bam_reader") != "samtools":
            # SAM text, or BAM decoded + name-collated natively (packed.bamChunks / packBam), packed natively
            pack = packAlignments(source, gk, keep_text=write_json)
            # the records cross PCIe in compact form (~30 bytes per mate instead of 128), like the bench's steps
            from .packed import CompactMates
            pack["compact"] = CompactMates(pack["records"], threads=2)
        return name, source, pack

    # the next sample is mapped / packed on a helper thread while this one is tabulated and written out
    # (three samples ahead on three threads: the serial stretches of one ingest leave cores to the others)
    ahead = max(1, int(testHook("ingest_ahead") or 3))       # samples packed ahead of the typing (the hook: tools/ingest_cli_sweep.sh)
    from concurrent.futures import ThreadPoolExecutor
    writer = ThreadPoolExecutor(max_workers=2, thread_name_prefix="gk-write")   # compact hand-off files, off this thread
    writes = []
    try:
        yield from _mapLoop(names, prepare, ahead, gk, gene_len, (copier, ingest), dindex, index_ref, exon_region_only,
                            write_json, writer, writes, keep_records)
    finally:
        for w in writes:
            w.result()          # every hand-off file is complete (and any write error surfaces) before we return
        writer.shutdown()


def _mapLoop(names, prepare, ahead, gk, gene_len, staging, dindex, index_ref, exon_region_only, write_json, writer, writes,
             keep_records=False):
    copier, dev = staging
    for name, source, pack in cohort.prefetched(range(len(names)), prepare, depth=ahead, workers=ahead):
        name += ".variant"
        logger.info(f"[Graph] Filter mapping ({name})")
        handed_off = False
        if pack is not None:
            # pinned records -> HBM on the copier's stream: compact words copied and expanded there, or the 128-byte records
            compact = pack.pop("compact", None)
            mates = compact.toDevice(copier, wait=True) if compact is not None else copier.put(pack["records"])
            data = extractVariantFromPacked(pack, gk, dev=dev, dindex=dindex, mates=mates)
            if compact is not None and not write_json and os.environ.get("GK_HANDOFF", "always") != "lazy":
                # the hand-off as the compact records that are in host memory anyway (70 MB for 2 M reads) instead of the
                # tabulated lists fetched back from the device (250 MB): hisat2.writeCompactRecords / loadCompact
                writes.append(writeCompactRecords(compact, pack, data.tab.novel_base, gk, name + ".npz",
                                                  index_ref=index_ref, background=writer))
                handed_off = True
        else:   # BAM name-collated through samtools like the reference (hisat2.readBam)
            data = extractVariant(readPair(source), gk, dev=dev, dindex=dindex)
        del pack
        if write_json:
            writeSampleJson(data, name + ".json")
            # the reference also rewrites the filtered pairs as BAM (hisat2.py:936-940)
            saveReadsToBam(data, name, source)
            saveReadsToBam(data, name + ".no_multi", source, filter_multi_mapped=True)
        elif handed_off:
            pass
        elif os.environ.get("GK_HANDOFF", "always") != "lazy":
            # compact hand-off instead: CSR + string table, no SAM text (hisat2.writeCompact).  GK_HANDOFF=lazy: only
            # for a sample that has to leave HBM before it is typed (main(): the --cn-cohort retention budget) -- a
            # cohort of 64 x 5 M reads would otherwise leave 42 GB of files that nothing reads
            writes.append(writeCompact(data, name + ".npz", index_ref=index_ref, background=writer))
        depth_name = name + ".no_multi"
        logger.info(f"[Graph] Calculate read depth to {depth_name}.depth.tsv")
        depthOfSample(data, gene_len, depth_name + ".depth.tsv", want_frame=False)
        depth_name += ".depth"
        if exon_region_only:
            logger.info(f"[Graph] Filter exon read to {depth_name}.exon.tsv")
            filterDepth(depth_name + ".tsv", depth_name + ".exon.tsv", readExons(index_ref))
            depth_name += ".exon"
        releaseInputs(data, keep_records)
        yield source, name, data, depth_name + ".tsv"


def releaseInputs(data: SampleData, keep_records: bool = False) -> None:
    """Drop what only the by-products and the depth needed: the SAM text of the pairs (host) and the packed
    records (HBM, 256 B per pair; unless ``keep_records``).  The tabulation's lists stay for typing."""
    data.pairs_text = None
    data._reads = None
    mates = getattr(data.tab, "mates", None)
    if mates is not None and not keep_records:
        mates.free()
        data.tab.mates = None


def readMapping(names, reads, index, index_ref, exon_region_only=False, alignments=None, write_json=True):
    """List form of ``mapSamples`` with the reference's return value (bam files, processed, depth files)."""
    bam_files, processed, depth_files = [], [], []
    for source, name, data, depth_file in mapSamples(names, reads, index, index_ref, exon_region_only,
                                                     alignments, write_json):
        bam_files.append(source)
        processed.append((name, data))
        depth_files.append(depth_file)
    return bam_files, processed, depth_files


def typingSuffix(name: str, cn_file: str, method: str) -> str:
    """``.cn<what the CN file's name adds to the sample's>.<method>`` (main.py:182-187)."""
    if method == "exonfirst":
        method += "_1"
    return ".cn" + cn_file[len(getCommonName(name, cn_file)):].replace("/", "_").replace(".", "_") + "." + method


def _plainField(x) -> bool:
    """A cell pandas' ``to_csv`` (tab separated, minimal quoting) writes as it is."""
    return isinstance(x, str) and not any(c in x for c in '\t"\n\r')


def writeTyping(name: str, typer, called_alleles: list[str], warning_genes: list[str]) -> str:
    """``{name}.tsv`` (name, alleles, warnings) and ``{name}.possible.tsv`` of one typed sample (main.py:203-219).

    The bytes are pandas' (``DataFrame.to_csv(sep="\\t", index=False)``, what the reference calls); for plain cells they
    are written directly -- the two frames cost 2.4 ms of interpreter time per sample, a third of the sample's GPU time
    with three typing lanes behind one interpreter lock -- anything else (a quote or a tab in a name, no rows) goes
    through pandas."""
    alleles, warnings = "_".join(called_alleles), "_".join(warning_genes)
    if all(_plainField(x) for x in (name, alleles, warnings)):
        with open(name + ".tsv", "w") as f:
            f.write(f"name\talleles\twarnings\n{name}\t{alleles}\t{warnings}\n")
    else:
        pd.DataFrame({"name": [name], "alleles": [alleles], "warnings": [warnings]}).to_csv(name + ".tsv", sep="\t", index=False)
    try:
        possible = typer.getAllPossibleTyping()       # reads the ranked results on the host only
    except NotImplementedError:      # EM strategy has no possible-set table (kir_typing.py:63-68)
        possible = []
    cols: list[str] = []
    for row in possible:                  # the frame's columns: keys in order of first appearance
        for k in row:
            if k not in cols:
                cols.append(k)
    plain = bool(possible) and cols[:3] == ["gene", "rank", "value"] and all(
        _plainField(r["gene"]) and isinstance(r["rank"], int) and isinstance(r["value"], (float, np.floating)) and
        np.isfinite(r["value"]) and all(_plainField(r[k]) and r[k] != "" for k in r if k not in ("gene", "rank", "value"))
        for r in possible)
    if plain:
        lines = ["\t".join(cols)]
        for r in possible:
            cells = [r["gene"], str(r["rank"]), repr(float(r["value"]))] + [r.get(k, "") for k in cols[3:]]
            lines.append("\t".join(cells))
        with open(name + ".possible.tsv", "w") as f:
            f.write("\n".join(lines) + "\n")
    else:
        pd.DataFrame(possible).fillna("").to_csv(name + ".possible.tsv", index=False, sep="\t")
    return name + ".tsv"


def sampleTyper(method: str, release: bool = True) -> "cohort.SampleTyper":
    """The typing stage of this process (``cohort.SampleTyper``: the sample lanes, search slots, urgent preamble and
    blocking waits that ``bench.py`` measures), finishing every sample the reference's way: its two files written, its
    tabulation released.  Submit ``(SampleData or hand-off file, copy numbers, (name, cn_file))``."""
    def finish(typer, called_alleles, warning_genes, item):
        name, cn_file, source = item
        if release or not isinstance(source, SampleData):      # done with this sample: free its HBM
            tab = typer._data.tab
            mates = getattr(tab, "mates", None)
            if mates is not None:
                mates.free()
                tab.mates = None
            tab.close()
        logger.info(f"[Allele] {called_alleles} ({name})")
        return writeTyping(name + typingSuffix(name, cn_file, method), typer, called_alleles, warning_genes)

    return cohort.SampleTyper(method, finish=finish)


def alleleTyping(processed_bam, cn_files: list[str], method: str = "full", release: bool = False) -> list[str]:
    """Allele typing of every sample; writes ``{name}{suffix}.tsv`` and ``.possible.tsv`` (171-220).

    ``processed_bam`` entries are names (the ``.json`` next to them is loaded) or (name, SampleData);
    ``release``: close every SampleData's tabulation once its results are taken (the pipeline does).
    The samples go through the process's typing lanes (``sampleTyper``): up to GK_SAMPLE_LANES at a time, results in
    order -- the reference types them one after the other (main.py:178-220), the files are the same."""
    allele_files = []
    with sampleTyper(method, release=release) as lanes:
        for entry, cn_file in zip(processed_bam, cn_files):
            name, source = entry if isinstance(entry, tuple) else (entry, entry + ".json")
            logger.debug(f"[Allele] Allele typing ({method}) with CN {cn_file} ({name})")
            lanes.submit(source, (lambda f=cn_file: loadCN(f)), (name, cn_file, source))
            if lanes.inFlight() >= lanes.lanes:
                allele_files.append(lanes.next())
        allele_files.extend(lanes.drain())
    return allele_files


def createParser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Run Graph-KIR (MI355X typing path)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--thread", default=1, help="Number of threads")
    p.add_argument("--engine", default="local", choices=["podman", "docker", "singularity", "local"],
                   help="How external tools (hisat2 / samtools) are run; only 'local' is supported here.")
    p.add_argument("--log-level", default="INFO", choices=logging._nameToLevel.keys(), help="Set the log level")
    p.add_argument("--r1", action="append", help="Paths to paired-end Read 1 FASTQ files")
    p.add_argument("--r2", action="append", help="Paths to paired-end Read 2 FASTQ files (in order)")
    p.add_argument("--input-csv", help="CSV with columns name, r1, r2 [, cnfile]")
    p.add_argument("--output-folder", help="Output folder (default: same folder as input reads)")
    p.add_argument("--output-cohort-name", help="Output prefix of the cohort files (default {output-folder}/cohort)")
    p.add_argument("--plot", action="store_true", help="(not supported in this build)")
    p.add_argument("--ipd-version", default="2100", help="IPD*KIR version")
    p.add_argument("--msa-type", default="ab_2dl1s1", choices=["merge", "split", "ab", "ab_2dl1s1"],
                   help="Type of MSA setup of the index")
    p.add_argument("--msa-no-exon-only-allele", action="store_true", help="Index without exon-only alleles")
    p.add_argument("--index-folder", default="index", help="Folder of the HISAT2-indexed KIR reference")
    p.add_argument("--index-wgs", help="(WGS extraction is outside this build)")
    p.add_argument("--ref-genome", default="hg19", choices=["hg19", "hg38"], help="(WGS extraction only)")
    p.add_argument("--cn-diploid-gene", choices=["", "VDR", "RYR1", "EGFR"], default="", help="(WGS extraction only)")
    p.add_argument("--cn-exon", action="store_true", help="Use exon-only depths for CN prediction")
    p.add_argument("--cn-cohort", action="store_true", help="Predict CN cohort-wise instead of sample-wise")
    p.add_argument("--cn-select", default="p75", choices=["p75", "mean", "median"], help="Gene depth statistic")
    p.add_argument("--cn-algorithm", default="LCND", choices=["LCND", "KDE"], help="CN prediction model")
    p.add_argument("--cn-dist-dev", default=0.08, help="Deviation of distributions in LCND")
    p.add_argument("--cn-3dl3-not-diploid", action="store_true", help="Do not assume KIR3DL3 is diploid")
    p.add_argument("--cn-provided", nargs="*", help="Provided CN TSVs (one per sample, in order)")
    p.add_argument("--allele-strategy", default="full", choices=["full", "pv", "exonfirst", "report", "em"],
                   help="full (alias pv): likelihood over all variants; exonfirst: exon variants first; "
                        "report (alias em): EM abundance typing")
    p.add_argument("--step-skip-extraction", action="store_true", help="Skip extracting KIR reads from WGS")
    p.add_argument("--step-skip-typing", action="store_true", help="Skip allele typing")
    # additive
    p.add_argument("--alignment", action="append",
                   help="Existing name-collated alignments (SAM / SAM.gz / BAM), one per sample: skips hisat2")
    p.add_argument("--no-variant-json", action="store_true", help="Do not write {name}.variant.json")
    p.add_argument("--ranks", type=int, default=1,
                   help="Start this many rank processes (samples are sharded over them; ranks map to GPUs round robin, "
                        "so 3 x the GPU count keeps every GPU busy).  Not needed under torchrun / any launcher that "
                        "sets RANK and WORLD_SIZE.")
    return p


def main(args: argparse.Namespace) -> None:
    if getattr(args, "cn_cohort", False):
        os.environ.setdefault("GK_SAMPLE_LANES", "3")      # samples wait in HBM for the pooled fit: the lanes' working sets stay small
    cohort.pipelineDefaults()       # before the first HIP call: blocking waits, sample lanes, search slots (as bench.py)
    setThreads(args.thread)
    setEngine(args.engine)
    logger.setLevel(args.log_level)
    logger.debug(f"[Main] {args}")
    if not args.step_skip_extraction and not args.alignment:
        raise NotImplementedError("WGS extraction (bwa) is outside this build: pass --step-skip-extraction")
    if args.plot:
        raise NotImplementedError("--plot is outside this build")

    if not args.input_csv:
        if not args.r1 and not args.alignment:
            raise ValueError("At least one paired-end read 1 FASTQ file must be provided")
        if args.r1:
            if len(args.r1) != len(args.r2 or []):
                raise ValueError("The number of paired-end read 1 and read 2 FASTQ files must be equal")
            reads = list(zip(args.r1, args.r2))
            names = [getCommonName(a, b) for a, b in reads]
        else:
            reads = [("", "")] * len(args.alignment)
            names = [a.rsplit(".", 2 if a.endswith(".gz") else 1)[0] for a in args.alignment]
        cn_files = list(args.cn_provided) if args.cn_provided else [""] * len(names)
    else:
        df = pd.read_csv(args.input_csv)
        names = list(df["name"])
        reads = list(zip(df["r1"], df["r2"]))
        cn_files = list(df["cnfile"].fillna("")) if "cnfile" in df.columns else [""] * len(names)
    if not names:
        raise ValueError("No samples found in input")
    if len(cn_files) != len(names):
        raise ValueError("Mismatch between number of samples and copy number files")
    if args.alignment and len(args.alignment) != len(names):
        raise ValueError("--alignment must be given once per sample")
    logger.info(f"[Main] Samples: {names}")

    if args.output_folder:
        Path(args.output_folder).mkdir(exist_ok=True)
        names = [replaceParentFolder(n, args.output_folder) for n in names]
        output_folder = args.output_folder
    else:
        output_folder = str(Path(names[0]).parent)
    cohort_name = args.output_cohort_name or str(Path(output_folder) / "cohort")
    Path(cohort_name).parent.mkdir(exist_ok=True)

    # index (must exist: building it needs the network, pyhlamsa and muscle -- out of scope)
    if args.msa_no_exon_only_allele:
        index_msa = f"{args.index_folder}/kir_{args.ipd_version}_{args.msa_type}.leftalign"
    else:
        index_msa = f"{args.index_folder}/kir_{args.ipd_version}_withexon_{args.msa_type}.leftalign"
    index_ref = index_msa + ".mut01"
    index = index_ref + ".graph"
    if not Path(index_ref + ".snp").exists():
        raise FileNotFoundError(f"{index_ref}.snp/.link/.locus not found: building the index is outside this build")

    # shard the cohort over ranks (one GPU per process): longest processing time first by input size
    transport = cohort.initFromEnv()
    comm = None
    if transport is not None:
        inputs = args.alignment if args.alignment else [list(r) for r in reads]
        comm = cohort.Comm(len(names), transport, weights=cohort.sampleWeights(inputs))
        try:
            return _runCohort(args, names, reads, cn_files, index, index_ref, cohort_name, comm)
        except BaseException as e:     # the other ranks must not wait out their timeouts for this one
            if transport.store is not None:
                transport.store.abort(f"rank {transport.rank}: {type(e).__name__}: {e}")
            raise
    return _runCohort(args, names, reads, cn_files, index, index_ref, cohort_name, comm)


def _runCohort(args, names, reads, cn_files, index, index_ref, cohort_name, comm) -> None:
    """Tabulation, depth, copy numbers, typing and the merges for this rank's share of the cohort (main.py:124-250,
    572-603)."""
    mine = comm.mine if comm else list(range(len(names)))
    pick = lambda xs: [xs[i] for i in mine]   # noqa: E731

    my_cn = pick(cn_files)
    kwargs = {"base_dev": float(args.cn_dist_dev), "start_base": 2}
    method = {"pv": "full", "report": "em"}.get(args.allele_strategy, args.allele_strategy)
    pooled_fit = args.cn_cohort and not all(cn_files)
    lanes = sampleTyper(method)
    try:
        allele_files, my_cn = _typeShare(args, lanes, pick(names), pick(reads), my_cn, pick, index, index_ref, cohort_name,
                                         comm, kwargs, pooled_fit)
    finally:
        lanes.close()
    _mergeShare(my_cn, allele_files, cohort_name, comm)


def _typeShare(args, lanes, names, reads, my_cn, pick, index, index_ref, cohort_name, comm, kwargs, pooled_fit):
    """Tabulation, depth, copy numbers of this rank's samples, each handed to the typing lanes as soon as its copy
    numbers are known; returns (allele files, CN files) in the rank's sample order."""
    allele_files: list[str] = []
    waiting: list[tuple[str, object]] = []      # samples that wait for the pooled copy-number fit
    depth_files: list[str] = []
    retained = 0
    budget = int(float(os.environ.get("GK_RETAIN_GB", "64")) * 2**30)   # tabulations kept in HBM until the pooled fit
    samples = mapSamples(names, reads, index, index_ref, exon_region_only=args.cn_exon,
                         alignments=pick(args.alignment) if args.alignment else None,
                         write_json=not args.no_variant_json, keep_records=pooled_fit and not args.step_skip_typing)
    for i, (_, name, data, depth_file) in enumerate(samples):
        depth_files.append(depth_file)
        if pooled_fit:
            # every sample's depth is needed before any can be typed.  What waits in HBM is the sample's packed records in
            # compact form (ParkedRecords: ~150 MB per 5 M reads), not its tabulation (~1 GB): the lists are made again when
            # its turn comes.  Beyond the budget a sample is parked in a hand-off file and reloaded for typing.
            if args.step_skip_typing:
                data.tab.close()
                waiting.append((name, None))
                continue
            size = int(0.12 * 256 * data.tab.n_pairs) + (1 << 20)      # what the compact records will take, roughly
            if retained + size > budget or data.tab.mates is None:
                if args.no_variant_json:
                    parked = name + ".npz"          # already written by mapSamples, unless hand-off files are lazy
                    if os.environ.get("GK_HANDOFF", "always") == "lazy":
                        writeCompact(data, parked, index_ref=index_ref)
                else:
                    parked = name + ".json"
                if data.tab.mates is not None:
                    data.tab.mates.free()
                    data.tab.mates = None
                data.tab.close()
                waiting.append((name, parked))
            else:
                parked = ParkedRecords(data)        # compacts the records, releases them and the tabulation
                retained += parked.nbytes
                waiting.append((name, parked))
            continue
        if not my_cn[i]:
            cn_name = str(Path(depth_file).with_suffix(f".{args.cn_select}.{args.cn_algorithm}"))
            logger.info(f"[CN] Copy number estimation per sample ({cn_name})")
            predictSamplesCN([depth_file], [cn_name + ".tsv"], "", cluster_method=args.cn_algorithm,
                             cluster_method_kwargs=kwargs, assume_3DL3_diploid=not args.cn_3dl3_not_diploid,
                             save_cn_model_path=cn_name + ".json", select_mode=args.cn_select)
            my_cn[i] = cn_name + ".tsv"
        # copy numbers known: hand the sample to the typing lanes and go on with the next one's tabulation; a lane
        # releases the sample when it is typed (at most GK_SAMPLE_LANES tabulations wait in HBM)
        if not args.step_skip_typing:
            lanes.submit(data, (lambda f=my_cn[i]: loadCN(f)), (name, my_cn[i], data))
            if lanes.inFlight() >= lanes.lanes:
                allele_files.append(lanes.next())
        else:
            data.tab.close()
    if pooled_fit:
        suffix = f".{args.cn_select}.cohort.{args.cn_algorithm}"
        my_cn = [str(Path(p).with_suffix(suffix + ".tsv")) for p in depth_files]
        logger.info(f"[CN] Copy number estimation by cohort ({cohort_name + suffix})")
        predictSamplesCN(depth_files, my_cn, cluster_method=args.cn_algorithm, cluster_method_kwargs=kwargs,
                         save_cn_model_path=cohort_name + suffix + ".json", select_mode=args.cn_select, comm=comm)
        for (name, source), cn_file in zip(waiting, my_cn):
            if args.step_skip_typing:
                continue
            if isinstance(source, ParkedRecords):
                source = source.restore()           # expansion + tabulation, while the lanes type the samples before it
            lanes.submit(source, (lambda f=cn_file: loadCN(f)), (name, cn_file, source))
            if lanes.inFlight() >= lanes.lanes:
                allele_files.append(lanes.next())
    allele_files.extend(lanes.drain())
    return allele_files, my_cn


def _mergeShare(my_cn, allele_files, cohort_name, comm) -> None:
    # merge on rank 0, in cohort order
    if comm is not None:
        cn_sorted = comm.gatherInCohortOrder(my_cn)
        al_sorted = comm.gatherInCohortOrder(allele_files)
        if comm.rank == 0:
            my_cn, allele_files = cn_sorted, [a for a in al_sorted if a]
    if comm is None or comm.rank == 0:
        logger.info(f"[CN] Saved copy number in {cohort_name}.cn.tsv")
        mergeCN(my_cn, cohort_name + ".cn.tsv")
        if allele_files:
            logger.info(f"[Allele] Saved in {cohort_name}.allele.tsv")
            mergeAllele(allele_files, cohort_name + ".allele.tsv")
    if comm is not None:
        comm.barrier()
        comm.close()
    logger.info("[Main] Success")


def spawnRanks(n: int, argv: list[str]) -> int:
    """``--ranks N`` without a launcher: start N copies of this command as ranks 0..N-1 (fresh processes; this
    one never touches the GPU) and supervise them: a rank that fails ends the launch for all (``comm.superviseRanks``),
    the rendezvous directory is removed either way.  Returns 0 or 1."""
    import subprocess
    import sys
    import tempfile
    import uuid
    from .comm import superviseRanks
    rdzv = tempfile.mkdtemp(prefix="gk_rdzv_")
    token = uuid.uuid4().hex
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   GK_RDZV_DIR=rdzv, GK_RDZV_TOKEN=token)
        procs.append(subprocess.Popen([sys.executable, "-m", "kir_graph_amd.main"] + argv, env=env))
    return superviseRanks(procs, rdzv, token)


def entrypoint() -> None:
    import sys
    args = createParser().parse_args()
    if args.ranks > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawnRanks(args.ranks, sys.argv[1:]))
    main(args)


if __name__ == "__main__":
    entrypoint()
