"""Host-side rehearsal of the 8-GPU node on ONE box: RANKS concurrent bench.py processes, each pinned to CORES_PER_RANK cores, each running the
N > 1 code path (DataParallel wrapper, phased hipGraph step, RCCL all-reduce through the C ABI communicator in a one-rank world:
`bench.py --dp-selftest`) on the one GPU.  Throughput is meaningless here (eight ranks share one GPU); what carries over to the 8-GPU node is the
HOST side of every rank, with eight ranks competing for the same 16 cores: the wall and CPU time the enqueueing thread needs to put ONE step into
empty queues (`bench.py --host-probe`: device synchronised before each probe step, so no back-pressure wait is counted) and its tail.  The
timed-loop numbers (thread / process CPU per step, per-step enqueue latency) are recorded too, but there the thread mostly WAITS for room in the
queues -- the runtime spins, so waiting shows up as CPU time in proportion to the (here eight times longer) device step; they are not the bar.
Writes gpurun_out/host8.json.

usage: python tools/host8.py [RANKS=8] [CORES_PER_RANK=2] [STEPS=12]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cpr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cores = sorted(os.sched_getaffinity(0))
assert ranks * cpr <= len(cores), f"{ranks} ranks x {cpr} cores need {ranks * cpr} cores; this box offers {len(cores)}"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


def solo_reference():
    """one rank alone on the box, same command: the uncontended host numbers"""
    return launch(0, 1)


def launch(i, n):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29610 + i), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               YTVLN_BENCH_RCCL_INFO="0", PYTORCH_HIP_ALLOC_CONF="expandable_segments:False")
    mine = cores[i * cpr:(i + 1) * cpr]
    cmd = ["taskset", "-c", ",".join(map(str, mine)), sys.executable, os.path.join(ROOT, "bench.py"), "--dp-selftest", "--no-variants",
           "--no-cpu-baseline", "--no-kernel-timing", "--steps", str(steps), "--warmup", "3", "--host-probe", "8"]
    err = open(os.path.join(ROOT, "gpurun_out", f"host8_rank{i}_of{n}.err"), "w")
    return subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=err, text=True), mine


def collect(procs):
    rows = []
    for i, (p, mine) in enumerate(procs):
        out, _ = p.communicate(timeout=1500)
        line = [x for x in out.splitlines() if x.startswith("{")]
        if p.returncode != 0 or not line:
            rows.append({"rank": i, "cores": mine, "error": f"rc={p.returncode}"})
            continue
        d = json.loads(line[-1])
        rows.append({"rank": i, "cores": mine, "ms_per_step": d["ms_per_step"], "host_cpu_ms_per_step": d["host_cpu_ms_per_step"],
                     "host_cpu_process_ms_per_step": d["host_cpu_process_ms_per_step"], "host_enqueue_ms_per_step": d["host_enqueue_ms_per_step"],
                     "host_enqueue_ms_tail": d["host_enqueue_ms_tail"], "host_probe": d["host_probe"], "execution": d["config"]["execution"],
                     "gradient_exchange": d["config"].get("gradient_exchange"), "hbm_reserved_gb": d["hbm_reserved_gb"]})
    return rows


t0 = time.time()
solo = collect([solo_reference()])[0]
t_solo = time.time() - t0
t0 = time.time()
rows = collect([launch(i, ranks) for i in range(ranks)])
ok = [r for r in rows if "error" not in r]
res = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "ranks": ranks, "cores_per_rank": cpr, "steps": steps, "host_cores": len(cores),
       "solo": solo, "per_rank": rows, "wall_s": {"solo": round(t_solo, 1), "concurrent": round(time.time() - t0, 1)}}
if ok and "error" not in solo:
    step_ms = solo["ms_per_step"]            # the step time each rank will have on its OWN GPU
    res["summary"] = {
        "step_ms_one_rank_alone": step_ms,
        "probe_enqueue_wall_ms": {"solo_p50": solo["host_probe"]["enqueue_wall_ms"]["p50"],
                                  "p50_max_over_ranks": max(r["host_probe"]["enqueue_wall_ms"]["p50"] for r in ok),
                                  "p50_mean_over_ranks": round(sum(r["host_probe"]["enqueue_wall_ms"]["p50"] for r in ok) / len(ok), 2),
                                  "max_over_ranks_and_steps": max(r["host_probe"]["enqueue_wall_ms"]["max"] for r in ok)},
        "probe_enqueue_cpu_ms": {"solo_p50": solo["host_probe"]["enqueue_cpu_ms"]["p50"],
                                 "p50_max_over_ranks": max(r["host_probe"]["enqueue_cpu_ms"]["p50"] for r in ok),
                                 "p50_mean_over_ranks": round(sum(r["host_probe"]["enqueue_cpu_ms"]["p50"] for r in ok) / len(ok), 2)},
        "probe_drain_ms": {"solo_p50": solo["host_probe"]["drain_ms"]["p50"], "p50_mean_over_ranks": round(sum(r["host_probe"]["drain_ms"]["p50"] for r in ok) / len(ok), 2)},
        "timed_loop_thread_cpu_ms_per_step": {"solo": solo["host_cpu_ms_per_step"], "max": max(r["host_cpu_ms_per_step"] for r in ok)},
        "timed_loop_process_cpu_ms_per_step": {"solo": solo["host_cpu_process_ms_per_step"], "max": max(r["host_cpu_process_ms_per_step"] for r in ok)},
        "bar": "per rank, 8 ranks on 16 cores: host work to enqueue one step (probe, wall, p50 and worst step) < 0.8 x the step time of one rank on its own GPU",
    }
    sm = res["summary"]
    sm["host_work_over_step"] = {"p50_max": round(sm["probe_enqueue_wall_ms"]["p50_max_over_ranks"] / step_ms, 3),
                                 "worst_step": round(sm["probe_enqueue_wall_ms"]["max_over_ranks_and_steps"] / step_ms, 3)}
    sm["pass"] = sm["host_work_over_step"]["p50_max"] < 0.8 and len(ok) == ranks
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "host8.json"), "w"), indent=1)
print(json.dumps(res.get("summary", res), indent=1))
