"""SR / SPL metrics (ytvln/metrics.py) against the reference's own `scripts/calculate-metrics.py` run on a synthetic data tree
(oracle/gen_golden_metrics.py -> tests/golden/g18_metrics.json: graphs, tasks, agent trajectories, and what the reference returned)."""
import json
import os

import pytest

from conftest import GOLD


def _tree(tmp_path, g):
    os.makedirs(tmp_path / "data" / "connectivity")
    os.makedirs(tmp_path / "data" / "task")
    for scan, d in g["connectivity"].items():
        json.dump(d, open(tmp_path / "data" / "connectivity" / f"{scan}_connectivity.json", "w"))
    json.dump(g["tasks"], open(tmp_path / "data" / "task" / "R2R_val_unseen.json", "w"))
    json.dump(g["results"], open(tmp_path / "results_val_unseen.json", "w"))


def test_metrics_match_the_reference_script(tmp_path):
    from ytvln import metrics
    g = json.load(open(os.path.join(GOLD, "g18_metrics.json")))
    _tree(tmp_path, g)
    ev = metrics.Evaluation(["val_unseen"], data_dir=str(tmp_path / "data"))
    summary, scores = ev.score(str(tmp_path / "results_val_unseen.json"))
    assert set(summary) == set(g["summary"])
    for k, v in g["summary"].items():
        assert abs(summary[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, summary[k], v)
    for k, ref in g["scores"].items():          # per-instruction values, in the order of the results file
        assert len(scores[k]) == len(ref) == 72
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(scores[k], ref)), k
    assert 0.0 < summary["spl"] <= summary["success_rate"] <= summary["oracle_success_rate"] <= 1.0


def test_metrics_error_behaviour(tmp_path, capsys):
    from ytvln import metrics
    g = json.load(open(os.path.join(GOLD, "g18_metrics.json")))
    _tree(tmp_path, g)
    ev = metrics.Evaluation(["val_unseen"], data_dir=str(tmp_path / "data"))
    with pytest.raises(AssertionError, match="Trajectories not provided"):          # a missing instruction id (calculate-metrics.py:143-146)
        ev.score_items(g["results"][1:])
    bad = [dict(r) for r in g["results"]]
    bad[0] = {"instr_id": bad[0]["instr_id"], "trajectory": [["vp_nowhere", 0, 0]]}
    with pytest.raises(AssertionError, match="start position"):                      # :104-106
        ev.score_items(bad)
    task = g["tasks"][0]
    adj = ev.graphs[task["scan"]]
    far = next(v for v in adj if v != task["path"][0] and v not in adj[task["path"][0]])
    bad[0] = {"instr_id": "%d_0" % task["path_id"], "trajectory": [[task["path"][0], 0, 0], [far, 0, 0]]}
    with pytest.raises(KeyError):                                                     # a jump along a non-edge (:117-128)
        ev.score_items(bad)
    metrics.main([str(tmp_path / "results_val_unseen.json"), "--data", str(tmp_path / "data")])
    printed = json.loads(capsys.readouterr().out)
    assert printed["success_rate"] == round(g["summary"]["success_rate"], 4)
    with pytest.raises(AssertionError):
        metrics.Evaluation(["val_other"], data_dir=str(tmp_path / "data"))
