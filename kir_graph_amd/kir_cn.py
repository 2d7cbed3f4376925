"""
Raw depths -> gene depths -> copy numbers -- drop-in for ``graphkir/kir_cn.py`` and
``graphkir/samtools_utils.readSamtoolsDepth``.

Same functions and arguments (``aggrDepths``, ``depthToCN``, ``predictSamplesCN``, ``loadCN``,
``filterDepth``, ``selectSamtoolsDepth``); the LCND grid search runs on the GPU (``cn_model.CNgroup``).

Cohort mode over several GPUs (``--cn-cohort``, main.py:572-589): the reference pools the raw gene
depths of ALL samples into one list and fits one model (kir_cn.py:61, 167-186; there is no
normalisation, SURVEY.md note 4).  With samples sharded over ranks, every rank contributes the
depths of its samples through ONE all-gather (``comm``), then runs the identical deterministic fit
on the identically ordered pool and assigns copy numbers to its own samples.
"""
from __future__ import annotations

import json
from itertools import chain
from typing import Any

import pandas as pd

from .cn_model import CNgroup, Dist, KDEcut
from .utils import logger


def readSamtoolsDepth(depth_filename: str) -> pd.DataFrame:
    """``samtools depth -aa`` TSV (gene, 1-based pos, depth), no header (samtools_utils.py:17-22)."""
    return pd.read_csv(depth_filename, sep="\t", header=None, names=["gene", "pos", "depth"])


def selectSamtoolsDepth(df: pd.DataFrame, ref_regions: dict[str, list[tuple[int, int]]]) -> pd.DataFrame:
    parts = []
    for gene, regions in ref_regions.items():
        for start, end in regions:
            parts.append(df[(df["gene"] == gene) & (start <= df["pos"]) & (df["pos"] <= end)])
    return pd.concat(parts)


def aggrDepths(depths: pd.DataFrame, select_mode: str = "p75") -> pd.DataFrame:
    """Per-gene depth: 75th percentile (linear interpolation) / mean / median over all positions."""
    grp = depths.groupby(by="gene", as_index=False)["depth"]
    if select_mode == "median":
        return grp.median()
    if select_mode == "mean":
        return grp.mean()
    if select_mode == "p75":
        return grp.quantile(0.75)
    raise NotImplementedError


def depthToCN(sample_gene_depths: list[dict[str, float]], diploid_depth: str = "", cluster_method: str = "CNgroup",
              cluster_method_kwargs: dict[str, Any] = {}, assume_3DL3_diploid: bool = False,
              pooled_values: list[float] | None = None) -> tuple[list[dict[str, int]], Dist]:
    """Fit one model to the pooled gene depths and assign a CN to every gene of every given sample.

    ``pooled_values``: depths of the whole cohort when the samples of this call are only a shard of
    it (multi-GPU cohort mode); defaults to the depths of ``sample_gene_depths``."""
    values = list(chain.from_iterable(s.values() for s in sample_gene_depths)) \
        if pooled_values is None else list(pooled_values)
    logger.info(f"[CN] Predict copy number by {cluster_method} with data size {len(values)}")
    if cluster_method == "CNgroup" or cluster_method.lower() == "lcnd":
        dist = CNgroup()
        if cluster_method_kwargs:
            dist = CNgroup.setParams(dist.getParams() | cluster_method_kwargs)
        lower_bound, upper_bound = 0.0, None
        if diploid_depth != "":
            with open(diploid_depth + ".json") as f:
                info = json.load(f)
            mean, dev = float(info["mean"]), float(info["std"])
            lower_bound, upper_bound = (mean - dev) / 2, (mean + dev) / 2
        else:
            dist.bin_num += 200
        dist.fit(values, lower_bound, upper_bound)
        if assume_3DL3_diploid:
            kir3dl3 = [float(s["KIR3DL3*BACKBONE"]) for s in sample_gene_depths]
            cn = dist.assignCN(kir3dl3)
            perc, rate, bins0 = float(1), 0.2, dist.bin_num
            while not all(c == 2 for c in cn):
                logger.debug("[CN] Assume 3DL3 cn=2")
                mid = sum(kir3dl3) / len(kir3dl3)
                dist.bin_num = int(bins0 * perc)
                dist.fit(values, (mid - perc * 10) / 2, (mid + perc * 10) / 2)
                cn = dist.assignCN(kir3dl3)
                perc = perc - rate
                if perc <= 0:
                    break
            assert all(c == 2 for c in cn)
        logger.info(f"[CN] {cluster_method} base = {dist.base}")
    elif cluster_method.lower() == "kde":
        dist = KDEcut()  # type: ignore[assignment]
        dist.fit(values)
        logger.info(f"[CN] {cluster_method} cut = {dist.local_min}")  # type: ignore[attr-defined]
    else:
        raise NotImplementedError
    out = []
    for s in sample_gene_depths:
        genes, depths = zip(*s.items())
        out.append(dict(zip(genes, dist.assignCN(depths))))  # type: ignore[arg-type]
    return out, dist


def filterDepth(depth_file: str, filtered_depth_file: str,
                bam_selected_regions: dict[str, list[tuple[int, int]]] = {}) -> None:
    depths = selectSamtoolsDepth(readSamtoolsDepth(depth_file), bam_selected_regions)
    depths.to_csv(filtered_depth_file, header=False, index=False, sep="\t")


def predictSamplesCN(samples_depth_tsv: list[str], samples_cn: list[str], diploid_depth: str = "",
                     save_cn_model_path: str | None = None, assume_3DL3_diploid: bool = False,
                     select_mode: str = "p75", per_gene: bool = False, cluster_method: str = "CNgroup",
                     cluster_method_kwargs: dict[str, Any] = {}, comm=None) -> None:
    """Depth TSVs -> per-sample CN TSVs (``gene, cn, depth``), one pooled fit (kir_cn.py:146-231).

    ``comm``: optional collective (``cohort.Comm``) when the listed samples are this rank's shard
    of a cohort; the depths of all ranks are all-gathered before the fit."""
    assert len(samples_depth_tsv) == len(samples_cn)
    tables = []
    for depth_file in samples_depth_tsv:
        logger.info(f"[CN] Select {select_mode} of depths per gene ({depth_file})")
        df = aggrDepths(readSamtoolsDepth(depth_file), select_mode=select_mode)
        df["depth_file"] = depth_file
        tables.append(df)
    logger.info(f"[CN] Predict CN from {len(tables)} samples")
    depths_dict = [dict(zip(t["gene"], t["depth"])) for t in tables]
    if per_gene:
        cns = _predictPerGene(tables, samples_depth_tsv, cluster_method, cluster_method_kwargs, save_cn_model_path, comm)
    else:
        pooled = None
        if comm is not None:
            pooled = comm.allgatherDepths(depths_dict)
        cns, model = depthToCN(depths_dict, diploid_depth, cluster_method=cluster_method,
                               cluster_method_kwargs=cluster_method_kwargs,
                               assume_3DL3_diploid=assume_3DL3_diploid, pooled_values=pooled)
        model.raw_df = [t.to_dict() for t in tables]
        if save_cn_model_path and (comm is None or comm.rank == 0):
            model.save(save_cn_model_path)
    for filename, cn, depths in zip(samples_cn, cns, depths_dict):
        df1 = pd.DataFrame(list(cn.items()), columns=["gene", "cn"])
        df2 = pd.DataFrame(list(depths.items()), columns=["gene", "depth"])
        df1.merge(df2, on="gene").to_csv(filename, index=False, sep="\t")


def _predictPerGene(tables: list[pd.DataFrame], samples_depth_tsv: list[str], cluster_method: str,
                    cluster_method_kwargs: dict[str, Any], save_cn_model_path: str | None, comm) -> list[dict[str, int]]:
    """``per_gene=True`` (kir_cn.py:195-222): one model per gene, fitted to that gene's depth in every sample -- no
    diploid-depth file, no 3DL3 assumption.  The reference keys the depths of one gene by ``gene + "-" + depth file``
    and maps a key back to its sample with ``key.split("-")[1]`` (line 217): a depth-file path that contains ``-``
    raises ``KeyError`` there, and so it does here."""
    from .utils import NumpyEncoder
    mine = None
    if comm is not None:
        # every gene is fitted over the WHOLE cohort: the per-gene depths of all ranks' samples are gathered in cohort
        # order (the order one process would read them in), every rank runs the same fits and keeps its own samples
        rows = comm.gatherInCohortOrder([(name, [str(g) for g in t["gene"]], [float(d) for d in t["depth"]])
                                         for name, t in zip(samples_depth_tsv, tables)])
        mine = [samples_depth_tsv.index(name) if name in samples_depth_tsv else -1 for name, _, _ in rows]
        tables = [pd.DataFrame({"gene": genes, "depth": depths, "depth_file": name}) for name, genes, depths in rows]
        samples_depth_tsv = [name for name, _, _ in rows]
        if comm.rank != 0:
            save_cn_model_path = None
    file_index = {name: i for i, name in enumerate(samples_depth_tsv)}
    df_depths = pd.concat(tables)
    df_depths["gene_sampleid"] = df_depths["gene"] + "-" + df_depths["depth_file"]
    cns: list[dict[str, int]] = [{} for _ in tables]
    models = []
    for gene in sorted(set(df_depths["gene"])):
        logger.info(f"[CN] Predict per gene: {gene}")
        gene_depths = df_depths[df_depths["gene"] == gene]
        gene_cns, gene_model = depthToCN([dict(zip(gene_depths["gene_sampleid"], gene_depths["depth"]))],
                                         cluster_method=cluster_method, cluster_method_kwargs=cluster_method_kwargs)
        gene_model.raw_df = [gene_depths.to_dict()]
        models.append((gene, gene_model))
        for gene_and_id, cn in gene_cns[0].items():
            cns[file_index[gene_and_id.split("-")[1]]][gene] = cn
    if save_cn_model_path:
        data = []
        for gene, model in models:
            data.append(model.getParams())
            data[-1]["gene"] = gene
            with open(save_cn_model_path + f".{gene}.json", "w") as f:
                json.dump(data[-1], f, cls=NumpyEncoder)
        with open(save_cn_model_path, "w") as f:
            json.dump(data, f, cls=NumpyEncoder)
    if mine is not None:          # this rank's samples, in the order they were given
        own = [(k, i) for i, k in enumerate(mine) if k >= 0]
        return [cns[i] for _, i in sorted(own)]
    return cns


def loadCN(filename_cn: str) -> dict[str, int]:
    """gene -> copy number of a CN file (``gene\\tcn[\\tdepth]``, kir_cn.py:234-243).  A plain file -- a header with a
    ``cn`` column, integer cells, no quotes -- is read directly (pandas' reader costs a millisecond of interpreter time
    per sample, which the typing lanes share); anything else goes through pandas like the reference."""
    try:
        with open(filename_cn) as f:
            text = f.read()
        if '"' not in text and "\r" not in text:
            lines = text.split("\n")
            if lines and lines[-1] == "":
                lines.pop()
            header = lines[0].split("\t")
            at = header.index("cn")
            out: dict[str, int] = {}
            for line in lines[1:]:
                cells = line.split("\t")
                if len(cells) != len(header) or not cells[0] or cells[0] in out:
                    raise ValueError
                out[cells[0]] = int(cells[at])
            if at > 0 and len(lines) > 1:
                return out
    except (ValueError, IndexError):
        pass
    data = pd.read_csv(filename_cn, sep="\t", index_col=[0])
    return dict(data.to_dict()["cn"])
