"""The hand-over into the reference's graph stage (SURVEY.md 8(b) "in-process consumer"; test infrastructure).

The reference's `graph` command does two things in one process: it produces the k-min-mer tables (the hot path rebuilt
here) and then builds the unitig graph from them -- createGfa() at k <= firstK+1, computeNextUnitigGraph() with the
in-memory table `_mdbgNodesLight` at k >= firstK+2 (graph/CreateMdbg.cpp:515-553, :3990, :4156, :5013).
`refdrv graph_from_tables` (oracle/ref_driver.cpp) runs that second half alone on tables found in the directory.

run_loop() drives the reference's multi-k loop (graph -> contig -> toMinspace per k, pipeline/AssemblyPipeline.hpp:603-671)
with a pluggable table producer; compare_dirs() checks that two such loops agree on everything the next stage reads.
"""
from __future__ import annotations

import dataclasses
import os
import shutil
import subprocess

import numpy as np

from metamdbg_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDRV = os.path.join(ROOT, "oracle", "_ref", "refdrv")

# what the graph stage writes (deterministic: unitigs are renamed by the hash of their content, CreateMdbg.cpp computeDeterministicUnitigs)
GRAPH_FILES = ("unitigGraph.nodes.bin", "unitigGraph.edges.successors.bin", "unitigGraph.nodes.abundances.bin", "unitigGraph.stats.bin")
# what contig / toMinspace leave for the next `graph`
NEXT_INPUTS = ("unitig_data.txt", "unitigGraph_prev.nodes.bin", "unitigGraph.nodes.refined_abundances.bin")
TABLE_FILES = ("kminmerData_abundance.txt", "kminmerData_min.txt")


def _run(cmd, timeout=600):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (cmd[:3], r.stderr[-1500:])


def graph_args(tmp: str, k: int, first_k: int, threads: int = 1) -> list[str]:
    return [tmp, "--threads", str(threads)] + (["--min-abundance", "0", "--firstpass"] if k == first_k else [])


def reference_graph(tmp: str, k: int, first_k: int) -> None:
    """The reference's whole `graph` command."""
    _run([REFDRV, "graph"] + graph_args(tmp, k, first_k))


def tables_then_reference_graph_stage(producer):
    """producer(tmp, k, first_k) writes the tables; the reference's own graph stage runs on top of them."""
    def step(tmp: str, k: int, first_k: int) -> None:
        producer(tmp, k, first_k)
        _run([REFDRV, "graph_from_tables"] + graph_args(tmp, k, first_k))
    return step


def run_loop(tmp: str, params: formats.Parameters, last_k: int, graph_step, snapshot_dir: str) -> None:
    """k = firstK .. last_k in `tmp` (read_data_corrected.txt, read_stats.txt already there); after every k the files of
    interest are copied to <snapshot_dir>/k<k>/."""
    first_k, prev_k = params.first_k, params.prev_k
    for k in range(first_k, last_k + 1):
        dataclasses.replace(params, kminmer_size=k, prev_k=prev_k, last_k=last_k).save(os.path.join(tmp, "parameters.gz"))
        graph_step(tmp, k, first_k)
        d = os.path.join(snapshot_dir, f"k{k}")
        os.makedirs(d, exist_ok=True)
        for name in GRAPH_FILES + TABLE_FILES:
            if os.path.exists(os.path.join(tmp, name)) and (name != "kminmerData_min.txt" or k <= first_k + 1):
                shutil.copy(os.path.join(tmp, name), os.path.join(d, name))
        sc = os.path.join(tmp, "smallContigs", f"smallContigs_k{k}.bin")
        if os.path.exists(sc):
            shutil.copy(sc, os.path.join(d, "smallContigs.bin"))
        if k == last_k:
            break
        _run([REFDRV, "contig", tmp, "--threads", "1", "--max-bubble-length", "50000", "--max-tip-length", "50000"])
        _run([REFDRV, "toMinspace", tmp, os.path.join(tmp, "contigs.nodepath"), os.path.join(tmp, "unitig_data.txt"),
              os.path.join(tmp, "unitigGraph.nodes.bin"), "--threads", "1"])
        for name in NEXT_INPUTS:
            shutil.copy(os.path.join(tmp, name), os.path.join(d, "next_" + name))
        prev_k = k


def compare_dirs(a: str, b: str, first_k: int, last_k: int) -> dict:
    """Every k: graph files and next-k inputs byte-equal, tables equal as multisets.  Returns counts for the assertion message."""
    from tests import multik_fixture as mk
    seen = {"graph_files": 0, "next_inputs": 0, "tables": 0}
    for k in range(first_k, last_k + 1):
        da, db = os.path.join(a, f"k{k}"), os.path.join(b, f"k{k}")
        rd = lambda d, n: open(os.path.join(d, n), "rb").read()
        for name in GRAPH_FILES:
            assert os.path.exists(os.path.join(da, name)) == os.path.exists(os.path.join(db, name)), (k, name)
            if os.path.exists(os.path.join(da, name)):
                assert rd(da, name) == rd(db, name), (k, name)
                seen["graph_files"] += 1
        assert np.array_equal(formats.sorted_abundance_records(rd(da, "kminmerData_abundance.txt")),
                              formats.sorted_abundance_records(rd(db, "kminmerData_abundance.txt"))), k
        seen["tables"] += 1
        if k <= first_k + 1:
            assert np.array_equal(formats.sorted_vector_records(rd(da, "kminmerData_min.txt"), k),
                                  formats.sorted_vector_records(rd(db, "kminmerData_min.txt"), k)), k
        if os.path.exists(os.path.join(da, "smallContigs.bin")):
            assert mk.small_contig_records(rd(da, "smallContigs.bin")) == mk.small_contig_records(rd(db, "smallContigs.bin")), k
        if k < last_k:
            for name in NEXT_INPUTS:
                assert rd(da, "next_" + name) == rd(db, "next_" + name), (k, name)
                seen["next_inputs"] += 1
    return seen


def shuffled_reference_tables(scratch: str, seed: int = 0):
    """A stand-in producer for CPU runs: the reference's own tables, records shuffled (another producer writes another
    order; the reference's own order already depends on its thread timing).  Tables are made by the reference's `graph` in a
    scratch copy of the directory, then only the table files travel."""
    rng = np.random.default_rng(seed)

    def producer(tmp: str, k: int, first_k: int) -> None:
        if os.path.exists(scratch):
            shutil.rmtree(scratch)
        shutil.copytree(tmp, scratch)
        _run([REFDRV, "graph"] + graph_args(scratch, k, first_k))
        ab = np.frombuffer(open(os.path.join(scratch, "kminmerData_abundance.txt"), "rb").read(), formats.ABUNDANCE_DTYPE)
        order = rng.permutation(len(ab))
        open(os.path.join(tmp, "kminmerData_abundance.txt"), "wb").write(ab[order].tobytes())
        if k <= first_k + 1:
            v = np.frombuffer(open(os.path.join(scratch, "kminmerData_min.txt"), "rb").read(), "<u4").reshape(-1, k)
            open(os.path.join(tmp, "kminmerData_min.txt"), "wb").write(v[order].tobytes())
        # the copies `graph` leaves next to the table (graph/CreateMdbg.cpp:515-522) and the small-contig file
        if k == first_k:
            shutil.copy(os.path.join(tmp, "kminmerData_abundance.txt"), os.path.join(tmp, "kminmerData_abundance_init.txt"))
        if k == first_k + 1:
            shutil.copy(os.path.join(tmp, "kminmerData_abundance.txt"), os.path.join(tmp, f"kminmerData_abundance_init_k{k}.txt"))
        shutil.copy(os.path.join(scratch, "smallContigs", f"smallContigs_k{k}.bin"), os.path.join(tmp, "smallContigs", f"smallContigs_k{k}.bin"))
    return producer
