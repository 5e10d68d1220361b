"""bench.py's host-side logic that needs no GPU."""
from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_deadline_ends_a_hung_run_with_a_line():
    """MDBG_BENCH_DEADLINE_S: a run that does not finish -- a collective whose peer never arrives; RCCL with more than one rank has not met
    hardware yet -- ends with ONE JSON line that carries "value": null and names the phase, the Python stacks on stderr, exit status 3."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench; bench._phase('waiting for a peer'); "
            "bench._arm_deadline(0, 8, os.dup(1)); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="1"))
    assert r.returncode == 3, r.stderr[-500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "waiting for a peer" in line["error"]
    assert "giving up" in r.stderr and "Thread" in r.stderr          # faulthandler's dump of every thread


def test_deadline_of_zero_never_fires_and_other_ranks_stay_silent_on_stdout():
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench; bench._arm_deadline(3, 8, os.dup(1)); time.sleep(2.5)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="0"))
    assert r.returncode == 0 and r.stdout == ""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25, env=dict(os.environ, MDBG_BENCH_DEADLINE_S="1"))
    assert r.returncode == 3 and r.stdout == "" and "rank 3 of 8" in r.stderr


def test_cpu_quota_is_read_from_the_cgroup():
    sys.path.insert(0, ROOT)
    import bench
    q = bench._cpu_quota()
    assert q is None or q > 0
    assert 1 <= bench._cores_used(32) <= 32


def _canned_result(k_blocks: int = 8, note_len: int = 600) -> dict:
    """A full result of the size a default run produces (per-k blocks, prose notes): what compact_line is given."""
    note = "x" * note_len
    per_k = {str(k): {"records_equal": True, "solid_equal": True, "abundance_checksum_equal": True, "sum_abundance_equal": True, "key_sum_equal": True,
                      "vector_sum_equal": True, "records": 11463338, "solid": 11463338, "abundance_checksum": 13619315941172088452, "note": note} for k in range(4, 4 + k_blocks)}
    check = {"all_equal": True, "reads": 10_000_000, "shards": 2, "k": list(range(4, 12)), "per_k": per_k, "seconds": 1.0, "mode": note}
    return {
        "metric": "Gbp/s through minimizer+k-min-mer step; bit-exact k-min-mer table vs ref", "value": 851.8184319585433, "unit": "Gbp/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 117.39591002988163, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": note, "workload_note": note, "reads_per_gpu": 10_000_000, "read_len": 10_000, "minimizers_per_step": 373753601,
                   "kminmer_records": 10775506, "solid": 10613396, "batches_in_flight": 2, "shared_device_options": {"scan_lds_reserve": 28672},
                   "overlap_probe": {"pairs": {"1": 20.5}}, "device": "gfx950:sramecc+:xnack-", "cus": 256,
                   "exchange": {"path": note, "rccl_ranks": 8, "ranks": 8, "wire_bytes_per_step": 1.0e9, "exchange_ms_per_step": 3.2, "gate": True, "note": note}},
        "roofline": {"bound": "valu", "kernel": note, "achieved": 245.49907819364225, "peak": 8000.0, "unit": "GB/s", "frac": 0.03068738477420528,
                     "traffic": 29846091980.8, "traffic_source": note, "bound_note": note, "algorithmic_bytes_per_launch": 28737536010.0,
                     "avg_launch_ms": 117.05761268615721, "note": note, "valu_floor": {"hash_cycles_per_64": 186, "floor_ms": 88.69, "frac": 0.7577}},
        "kernel_ms_per_step": {"scan": 117.05, "scan_compact": 0.0, "purge_palindromes": 2.94, "kminmer_split": 21.27, "kminmer_insert": 24.2},
        "cpu_baseline": {"value": 0.2345835773155646, "unit": "Gbp/s", "cores": 16, "threads": 32, "cpu_quota": 16.0, "kind": "reference", "sample": note,
                         "path_only": {"read_selection_s": 23.1}, "whole_commands": {"note": note}},
        "roofline_kminmer": {"bound": "hbm", "kernel": note, "achieved": 443.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.0554, "traffic": 44954009668.2,
                             "traffic_over_algorithmic": 6.23, "algorithmic_bytes": 7210578340.0, "kernel_ms_total": 16.27, "note": note, "one_table_pass": {"note": note}},
        "self_check": check,
        "roofline_index": {"per_k": {str(k): {"bound": "hbm", "achieved": 340.0, "frac": 0.0425, "kernel_ms_total": 19.2345, "traffic": 2.0e10, "note": note} for k in range(4, 12)}},
        "roofline_ont": {"scan": {"note": note}, "kminmer": {"note": note}},
        "parity": {"reads": 1_000_000, "bases": 10 ** 10, "init_bytes_equal": True, "corrected_multiset_equal": True, "table_multiset_equal": True,
                   "abundance_checksum_equal": True, "abundance_checksum": 17724130310159059115, "against": note, "golden": {"fixture": "x", "digests_equal": True}},
        "legs": {"end_to_end": {"workload": note, "mdbg_tool_gbps": 9.55, "init_bytes_equal": True},
                 "multik": {"seconds": 0.2722992890048772, "gbps": 367.2, "roofline_per_k": {"4": {"note": note}}, "self_check": check},
                 "multik_reference": {"workload": note, "k_done": list(range(4, 12)), "complete": True, "all_tables_equal": True, "per_k": per_k},
                 "pcie": {"workload": note, "packed_one_context_pipelined_gbps": 199.7},
                 "ont": {"gbps": 382.7, "parity": {"reads": 100_000, "init_bytes_equal": True, "table_multiset_equal": True}, "self_check": check,
                         "roofline": {"scan": {"note": note}}, "workload": note}},
        "speedup_vs_cpu_reference_path_only": 3631.19,
    }


def test_the_stdout_line_is_compact_and_complete():
    """The driver parses ONE line out of an 8 KB tail (round 4's 28 KB line left `parsed: null`): whatever the legs produce, the line stays
    under 4 KB and carries the contract's keys, both rooflines, the CPU baseline, the parity booleans, the checks and one number per leg."""
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_result()
    assert len(json.dumps(full)) > 20_000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096
    assert "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "roofline_kminmer", "cpu_baseline", "parity", "checks", "legs"):
        assert key in line, key
    assert line["value"] == 851.818 and line["ms_per_step"] == 117.396 and line["steps"] == 20 and line["warmup"] == 5
    assert set(line["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "valu_floor"}
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-6
    assert set(line["roofline_kminmer"]) >= {"bound", "kernel", "achieved", "peak", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch", "avg_launch_ms"}
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "threads", "kind", "sample"} and len(line["cpu_baseline"]["sample"]) <= 200
    assert len(line["config"]["workload"]) <= 200 and line["config"]["exchange"]["rccl_ranks"] == 8
    assert line["parity"] == {"init_bytes_equal": True, "corrected_multiset_equal": True, "table_multiset_equal": True, "abundance_checksum_equal": True,
                              "reads": 1_000_000, "against": line["parity"]["against"], "golden_digests_equal": True}
    assert line["checks"] == {"self_check": True, "multik_self_check": True, "multik_reference": True, "ont_parity": True, "ont_self_check": True}
    assert line["legs"] == {"multik_s": 0.272299, "multik_gbps": 367.2, "ont_gbps": 382.7, "pcie_gbps": 199.7, "e2e_gbps": 9.55}
    assert line["roofline_index"]["6"] == {"ms": 19.23, "frac": 0.0425, "traffic": 2.0e10} and sorted(line["roofline_index"], key=int) == [str(k) for k in range(4, 12)]


def test_the_stdout_line_reports_broken_and_absent_legs():
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_result()
    full["legs"]["ont"] = {"error": "MemoryError: no room"}
    del full["legs"]["multik_reference"]
    full["self_check"]["all_equal"] = False
    line = bench.compact_line(full)
    assert line["checks"]["self_check"] is False and line["checks"]["ont_parity"] is False and line["checks"]["ont_self_check"] is False
    assert line["checks"]["multik_reference"] is None and line["legs"]["errors"] == ["ont"] and "ont_gbps" not in line["legs"]
    # an N > 1 line: no legs, no CPU baseline, a parity block of its own
    multi = {k: v for k, v in _canned_result().items() if k not in ("legs", "cpu_baseline", "roofline_kminmer", "self_check")}
    multi.update(n_gpus=8, cpu_baseline=None, parity={"reads": 40_000_000, "table_equal": True, "records_equal": True, "mode": "x" * 500, "single_gpu_gbps": 790.123456})
    line = bench.compact_line(multi)
    assert len(json.dumps(line)) < 4096 and line["cpu_baseline"] is None and line["parity"]["table_equal"] is True and line["parity"]["single_gpu_gbps"] == 790.123


def test_emit_writes_the_detail_file_and_one_line(tmp_path):
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; bench.ROOT = %r; "
            "from tests.test_bench_host_logic import _canned_result; bench.emit(_canned_result(), os.dup(1))" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-500:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["detail"] == "bench_detail.json"
    detail = json.load(open(tmp_path / "bench_detail.json"))
    assert detail["legs"]["multik"]["self_check"]["per_k"]["4"]["records"] == 11463338
    assert "[bench] full result: {" in r.stderr


def test_counter_traffic_is_reported_only_for_the_sources_it_was_collected_on(tmp_path, monkeypatch):
    """roofline_index.per_k.*.traffic comes from profiles/*_index_traffic.json (tools/index_traffic.sh) and only when the file's git blob
    hashes are those of the tree's csrc/kminmer.hip, table.hpp and kminmer_dev.hpp: a collection made on another version of the kernels is
    not reported (None, with the reason)."""
    sys.path.insert(0, ROOT)
    import bench
    here = {f: bench.git_blob_hash(os.path.join(ROOT, "metamdbg_amd", "csrc", f)) for f in ("kminmer.hip", "table.hpp", "kminmer_dev.hpp")}
    per_kernel = {"_ZN4mdbg23prev_abundance_u_kernelILi1ELb1EEE": {"traffic_bytes_uncorrected": 26.0e9}, "_ZN4mdbg21index_insert_u_kernelILi1ELb1EEE": {"traffic_bytes_uncorrected": 20.0e9},
                  "_ZN4mdbg24distinct_insert_u_kernelILi1ELb1EEE": {"traffic_bytes_uncorrected": 30.0e9}, "_ZN4mdbg19refine_slots_kernelE": {"traffic_bytes_uncorrected": 5.0e9},
                  "_ZN4mdbg16slot_flag_kernelE": {"traffic_bytes_uncorrected": 1.9e9}}
    prof = tmp_path / "profiles"
    prof.mkdir()
    import bench_legs
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench_legs, "ROOT", str(tmp_path))          # (the look-up of committed PMC collections lives there since round 6)
    os.makedirs(tmp_path / "metamdbg_amd" / "csrc")
    for f in here:
        (tmp_path / "metamdbg_amd" / "csrc" / f).write_bytes(open(os.path.join(ROOT, "metamdbg_amd", "csrc", f), "rb").read())
    assert bench.index_traffic(10_000_000, 10_000) == (None, "no PMC collection for this workload under profiles/")
    (prof / "x_index_traffic.json").write_text(json.dumps({"reads": 10_000_000, "read_len": 10_000, "blobs": dict(here, **{"kminmer.hip": "0" * 40}), "per_kernel": per_kernel}))
    t, note = bench.index_traffic(10_000_000, 10_000)
    assert t is None and "another version" in note
    (prof / "y_index_traffic.json").write_text(json.dumps({"reads": 10_000_000, "read_len": 10_000, "blobs": here, "per_kernel": per_kernel}))
    t, note = bench.index_traffic(10_000_000, 10_000)
    assert t == {"refined": 35.0e9, "index": 46.0e9} and "y_index_traffic.json" in note
    assert bench.index_traffic(1_000_000, 10_000)[0] is None
