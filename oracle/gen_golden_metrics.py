"""Golden for the SR / SPL metrics (scripts/calculate-metrics.py of the reference).  TEST INFRASTRUCTURE ONLY; build container only: writes a
synthetic data/ tree (two scans with random 3-D poses and a random undirected visibility graph incl. excluded viewpoints, R2R-style tasks whose
reference paths are graph walks, agent trajectories that reach / miss / overshoot the goal and repeat viewpoints), runs the REAL reference
`Evaluation` on it (needs networkx, present here) and stores inputs + what it returned.

    python oracle/gen_golden_metrics.py        -> tests/golden/g18_metrics.json
"""
import importlib.util
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def make_scan(rng, n):
    pts = [(rng.uniform(0, 12), rng.uniform(0, 12), rng.uniform(0, 3)) for _ in range(n)]
    included = [rng.random() > 0.1 for _ in range(n)]
    vis = [[False] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            d = sum((a - b) ** 2 for a, b in zip(pts[i], pts[j])) ** 0.5
            if d < 4.0 and rng.random() < 0.8:
                vis[i][j] = vis[j][i] = True
    data = []
    for i in range(n):
        pose = [0.0] * 16
        pose[3], pose[7], pose[11] = pts[i]
        data.append({"image_id": "vp%03d" % i, "pose": pose, "included": included[i], "unobstructed": vis[i]})
    return data


def walk(rng, data, steps):
    idx = {d["image_id"]: i for i, d in enumerate(data)}
    nbr = lambda v: [data[j]["image_id"] for j, c in enumerate(data[idx[v]]["unobstructed"]) if c and data[j]["included"]]      # noqa: E731
    starts = [d["image_id"] for d in data if d["included"] and nbr(d["image_id"])]
    p = [rng.choice(starts)]
    for _ in range(steps):
        p.append(rng.choice(nbr(p[-1])))
    return p


def main():
    rng = random.Random(7)
    scans = {"scanA": make_scan(rng, 40), "scanB": make_scan(rng, 28)}
    tasks, results = [], []
    for pid in range(24):
        scan = rng.choice(sorted(scans))
        path = walk(rng, scans[scan], rng.randint(3, 6))
        tasks.append({"path_id": pid, "scan": scan, "path": path, "heading": 0.0, "instructions": ["a", "b", "c"]})
        for k in range(3):
            kind = rng.random()
            traj = list(path) if kind < 0.4 else list(path[: rng.randint(1, len(path))])
            if kind > 0.7:                      # wander on from wherever it stopped
                data = scans[scan]
                idx = {d["image_id"]: i for i, d in enumerate(data)}
                for _ in range(rng.randint(1, 4)):
                    nb = [data[j]["image_id"] for j, c in enumerate(data[idx[traj[-1]]]["unobstructed"]) if c and data[j]["included"]]
                    traj.append(rng.choice(nb))
            if rng.random() < 0.3:              # a repeated viewpoint (the agent turning in place)
                at = rng.randrange(len(traj))
                traj.insert(at, traj[at])
            results.append({"instr_id": "%d_%d" % (pid, k), "trajectory": [[v, 0.0, 0.0] for v in traj]})
    results.append({"instr_id": "999_0", "trajectory": [["vp000", 0, 0]]})          # an id the split does not have: ignored
    spec = importlib.util.spec_from_file_location("calc_metrics", "/root/reference/scripts/calculate-metrics.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "connectivity")); os.makedirs(os.path.join(tmp, "data", "task"))
        for s, d in scans.items():
            json.dump(d, open(os.path.join(tmp, "data", "connectivity", s + "_connectivity.json"), "w"))
        json.dump(tasks, open(os.path.join(tmp, "data", "task", "R2R_val_unseen.json"), "w"))
        json.dump(results, open(os.path.join(tmp, "results_val_unseen.json"), "w"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            summary, scores = ref.Evaluation(["val_unseen"]).score("results_val_unseen.json")
        finally:
            os.chdir(cwd)
    out = {"connectivity": scans, "tasks": tasks, "results": results, "summary": {k: float(v) for k, v in summary.items()},
           "scores": {k: [float(x) for x in v] for k, v in scores.items()}}
    json.dump(out, open(os.path.join(GOLD, "g18_metrics.json"), "w"))
    print("g18 ok", out["summary"])


if __name__ == "__main__":
    main()
