"""Inspection reports and training figures (host side): the rows and printed lines of the
reference's check_pt.py / scripts_check.py as captured in tests/golden/report_cases.json (generated
by tests/golden/make_golden_report.py from the reference scripts themselves), and the files the
metrics logger writes."""
import csv
import json
import os

import numpy as np
import pytest
import torch

from gan_amd.data import Data
from gan_amd.export import convert_graphml_to_bench, pt_to_graphml
from gan_amd.report import (CSV_FIELDS, check_bench_lines, check_graphml_lines, depth_diagnosis, pt_parameters,
                            write_pt_parameters_csv)

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "report_cases.json"), encoding="utf-8") as _f:
    CASES = json.load(_f)


def graph_of(key):
    name, tag = key.split("|")
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    pre = "enf_" if tag == "enf" else ""
    return {k: torch.from_numpy(np.asarray(z[pre + k])) for k in ("x", "edge_index", "edge_attr", "node_depth")}


@pytest.mark.parametrize("key", sorted(CASES))
def test_reports_match_reference_scripts(key, tmp_path):
    g = graph_of(key)
    want = CASES[key]
    d = Data(**g)
    assert pt_parameters(d, "g.pt") == want["row"]
    assert depth_diagnosis(d) == want["depth_lines"]
    pt, gml, bench = (str(tmp_path / f"g.{ext}") for ext in ("pt", "graphml", "bench"))
    torch.save(g, pt)
    pt_to_graphml(pt, gml, verbose=False)
    convert_graphml_to_bench(gml, bench)
    got = [ln.replace(str(tmp_path), "<tmp>") for ln in check_graphml_lines(gml) + check_bench_lines(bench)]
    assert got == want["check_lines"]


def test_parameter_csv_and_error_rows(tmp_path, capsys):
    keys = sorted(CASES)[:3]
    for i, key in enumerate(keys):
        torch.save(graph_of(key), str(tmp_path / f"s{i}.pt"))
    torch.save({"x": torch.zeros(3, 2), "edge_index": torch.zeros(2, 0, dtype=torch.long),
                "edge_attr": torch.zeros(0, 1)}, str(tmp_path / "nodepth.pt"))      # no depth attribute: an error row
    out_csv = str(tmp_path / "params.csv")
    rows = write_pt_parameters_csv(str(tmp_path), out_csv, verbose=False)
    with open(out_csv, newline="", encoding="utf-8") as f:
        read = list(csv.DictReader(f))
    assert list(read[0].keys()) == CSV_FIELDS and len(read) == len(rows) == 4
    by_name = {r["filename"]: r for r in read}
    for i, key in enumerate(keys):
        want = CASES[key]["row"]
        assert {k: by_name[f"s{i}.pt"][k] for k in ("PI", "PO", "N", "E", "I", "D", "status")} == \
               {k: str(want[k]) for k in ("PI", "PO", "N", "E", "I", "D", "status")}
    assert by_name["nodepth.pt"]["status"].startswith("error: ") and by_name["nodepth.pt"]["PI"] == ""
    empty = tmp_path / "empty"
    empty.mkdir()
    assert write_pt_parameters_csv(str(empty), out_csv) == []
    assert "未找到任何.pt文件" in capsys.readouterr().out


def test_metrics_logger_writes_csv_and_figures(tmp_path):
    from gan_amd.train import METRIC_FIELDS, PLOTS, MetricsLogger
    lg = MetricsLogger(str(tmp_path))
    assert lg.plot_all() == []
    for s in range(5):
        lg.log(s, 0, d_real=0.1 * s, d_fake=-0.2 * s, gap=0.3 * s, grad_norm=1.0 + s, loss_d=2.0 - s, loss_g=0.5 * s,
               gp=0.01 * s, pen=3.0)
    lg.flush_csv()
    with open(tmp_path / "metrics.csv", newline="") as f:
        rows = list(csv.DictReader(f))
    assert list(rows[0].keys()) == METRIC_FIELDS and len(rows) == 5 and float(rows[4]["gap"]) == pytest.approx(1.2)
    pytest.importorskip("matplotlib")
    written = lg.plot_all()
    assert sorted(os.path.basename(p) for p in written) == sorted(PLOTS) == ["gap_grad.png", "gp_pen.png", "losses.png", "scores.png"]
    for p in written:
        with open(p, "rb") as f:
            assert f.read(8) == b"\x89PNG\r\n\x1a\n"
