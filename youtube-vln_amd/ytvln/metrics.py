"""Standard VLN path metrics (navigation error, oracle success, success rate, SPL) of the inference re-ranking path
(`scripts/calculate-metrics.py:1-202` of the reference).  Host code: the graphs are a few hundred
viewpoints per scan, so all-pairs shortest paths are a binary-heap Dijkstra per node -- no graph library.

    ev = Evaluation(["val_unseen"], data_dir="data")             # data/connectivity/<scan>_connectivity.json, data/task/R2R_<split>.json
    summary, scores = ev.score("results/val_unseen.json")        # or ev.score_items(list_of_results)

Same definitions as the reference: success = final viewpoint within 3 m (geodesic) of the goal, oracle success = ANY visited
viewpoint within 3 m, trajectory length = sum of geodesic distances between consecutive distinct viewpoints (which must be
graph neighbours), SPL = success x shortest / max(length, shortest), averages over all instruction ids of the split (three
per path).
"""
from __future__ import annotations

import heapq
import json
import os
from collections import defaultdict
from typing import Dict, Iterable, List, Sequence, Tuple

ERROR_MARGIN = 3.0          # calculate-metrics.py:72


def load_nav_graph(path: str) -> Dict[str, Dict[str, float]]:
    """Adjacency {viewpoint: {neighbour: metres}} of one scan (calculate-metrics.py:14-48): included viewpoints joined where
    `unobstructed`, weighted by the Euclidean distance of their poses (elements 3, 7, 11 of the 4x4 pose)."""
    with open(path) as f:
        data = json.load(f)
    adj: Dict[str, Dict[str, float]] = {}
    for i, item in enumerate(data):
        if not item["included"]:
            continue
        for j, conn in enumerate(item["unobstructed"]):
            if conn and data[j]["included"]:
                if not data[j]["unobstructed"][i]:
                    raise AssertionError("Graph should be undirected")
                a, b = item["pose"], data[j]["pose"]
                w = ((a[3] - b[3]) ** 2 + (a[7] - b[7]) ** 2 + (a[11] - b[11]) ** 2) ** 0.5
                adj.setdefault(item["image_id"], {})[data[j]["image_id"]] = w
                adj.setdefault(data[j]["image_id"], {})[item["image_id"]] = w
    return adj


def all_pairs_shortest(adj: Dict[str, Dict[str, float]]) -> Dict[str, Dict[str, float]]:
    """{source: {reachable node: geodesic metres}} (the reference's `nx.all_pairs_dijkstra_path_length`)."""
    out = {}
    for src in adj:
        dist = {src: 0.0}
        heap: List[Tuple[float, str]] = [(0.0, src)]
        while heap:
            d, u = heapq.heappop(heap)
            if d > dist[u]:
                continue
            for v, w in adj[u].items():
                nd = d + w
                if nd < dist.get(v, float("inf")):
                    dist[v] = nd
                    heapq.heappush(heap, (nd, v))
        out[src] = dist
    return out


class Evaluation:
    """calculate-metrics.py:60-187.  `data_dir` holds `connectivity/` and `task/` (the reference hard-codes "data")."""

    def __init__(self, splits: Sequence[str], data_dir: str = "data"):
        self.error_margin = ERROR_MARGIN
        self.splits = list(splits)
        self.gt, ids, scans = {}, [], []
        for split in self.splits:
            if split not in ("train", "val_seen", "val_unseen", "test"):
                raise AssertionError(split)
            with open(os.path.join(data_dir, "task", f"R2R_{split}.json")) as f:
                for item in json.load(f):
                    self.gt[item["path_id"]] = item
                    scans.append(item["scan"])
                    ids += ["%d_%d" % (item["path_id"], i) for i in range(3)]
        self.scans, self.instr_ids = set(scans), set(ids)
        self.graphs = {s: load_nav_graph(os.path.join(data_dir, "connectivity", f"{s}_connectivity.json")) for s in self.scans}
        self.distances = {s: all_pairs_shortest(g) for s, g in self.graphs.items()}

    def _score_item(self, scores, instr_id: str, path) -> None:
        gt = self.gt[int(instr_id.split("_")[0])]
        dist, graph = self.distances[gt["scan"]], self.graphs[gt["scan"]]
        start, goal = gt["path"][0], gt["path"][-1]
        if start != path[0][0]:
            raise AssertionError("Result trajectories should include the start position")
        near_d = dist[path[0][0]][goal]
        for item in path:                                   # oracle stopping rule: the closest visited viewpoint
            near_d = min(near_d, dist[item[0]][goal])
        scores["nav_errors"].append(dist[path[-1][0]][goal])
        scores["oracle_errors"].append(near_d)
        length, prev = 0.0, path[0]
        for curr in path[1:]:
            if prev[0] != curr[0] and curr[0] not in graph[prev[0]]:
                raise KeyError(f"trajectory moves from {prev[0]} to {curr[0]} but the navigation graph has no such edge")
            length += dist[prev[0]][curr[0]]
            prev = curr
        scores["trajectory_lengths"].append(length)
        scores["shortest_path_lengths"].append(dist[start][goal])

    def score_items(self, results: Iterable[dict]):
        scores = defaultdict(list)
        todo = set(self.instr_ids)
        for item in results:
            if item["instr_id"] in todo:
                todo.remove(item["instr_id"])
                self._score_item(scores, item["instr_id"], item["trajectory"])
        if todo:
            raise AssertionError("Trajectories not provided for %d instruction ids: %s" % (len(todo), todo))
        n = len(scores["nav_errors"])
        spls = [sp / max(length, sp) if err < self.error_margin else 0.0
                for err, length, sp in zip(scores["nav_errors"], scores["trajectory_lengths"], scores["shortest_path_lengths"])]
        summary = {
            "length": sum(scores["trajectory_lengths"]) / n,
            "nav_error": sum(scores["nav_errors"]) / n,
            "oracle_success_rate": sum(e < self.error_margin for e in scores["oracle_errors"]) / n,
            "success_rate": sum(e < self.error_margin for e in scores["nav_errors"]) / n,
            "spl": sum(spls) / n,
        }
        assert summary["spl"] <= summary["success_rate"] + 1e-12
        return summary, dict(scores)

    def score(self, output_file: str):
        with open(output_file) as f:
            return self.score_items(json.load(f))


def main(argv=None) -> None:
    """`python -m ytvln.metrics results/val_unseen.json [--data data]` prints what the reference script prints (4 decimals)."""
    import argparse
    ap = argparse.ArgumentParser("Calculate standard VLN metrics")
    ap.add_argument("path")
    ap.add_argument("--data", default="data")
    a = ap.parse_args(argv)
    split = "val_unseen" if "val_unseen" in a.path else "val_seen"          # calculate-metrics.py:194
    summary, _ = Evaluation([split], a.data).score(a.path)
    print(json.dumps({k: round(v, 4) for k, v in summary.items()}, indent=2))


if __name__ == "__main__":
    main()
