#!/usr/bin/env python
"""Headline benchmark: ViLBERT pre-training throughput in traj-instr pairs/s on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

Workload (N = 1 ... 8, weak scaling): BASELINE.json configs[1] per GPU -- full ViLBERT (12 text / 6 image / 6 co-attention
layers, 250 M parameters), bs = 8 items x K = 7 options = 56 pairs, 80 tokens x (8 frames x 36 regions) x 2048-d, fp32,
train mode (dropout on), one step = forward + MLM/MVM/ranking/traj losses + backward + (all-reduce) + fused AdamW + LR step,
exactly the body of the reference's train_epoch (utils/utils_init.py:199-239).  Inputs are synthetic and resident in HBM.

Prints ONE JSON line (rank 0).  Besides the driver contract it carries
  roofline     -- the dominant kernel (fp32 MFMA GEMM): algorithmic FLOPs of its launches / their HIP-event time, measured
                  live on the launch stream during the timed steps, against the 157.3 TFLOP/s fp32 matrix peak;
  variants     -- (N = 1, default precision only; NOT the headline) the same captured step re-timed with the opt-in fp32x3 projections
                  (fp32 operands, three exact bf16 terms per value, six bf16 MFMAs per product; LABNOTES.md 5a) and with bf16 operands
                  (the arithmetic of BASELINE configs[4]); --no-variants skips them;
  cpu_baseline -- the CPU oracle (a port of the reference's path, oracle/vilbert_ref.py) timed on this box's host cores
                  on a bounded sample of the same workload (rank 0, N = 1 only).

N > 1: `python bench.py --gpus N` starts its own N ranks (re-exec under torch.distributed.run on 127.0.0.1) unless a launcher already
exported WORLD_SIZE; one rank per GPU, gradients summed over RCCL/xGMI through the C ABI's communicator (ytvln_rccl_*).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver (multi-process GPU runs)
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "youtube-vln_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (no sparsity)
TRAIN_GFLOP_PER_PAIR = 223.9   # algorithmic, T=80 R=288 full config, training = 3 x forward (SURVEY.md 8d / BASELINE.md 3)

WORKLOADS = {
    # name: (config json, bs, K, T, frames, boxes, flags)
    "cfg2_full_pretrain_bs8": ("bert_base_6_layer_6_connect.json", 8, 7, 80, 8, 36, dict(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)),
    # the other reading of "bs=8" -- 8 model rows (K = 1 option per item), same losses; a latency-sized launch set
    "cfg2_k1_rows8": ("bert_base_6_layer_6_connect.json", 8, 1, 80, 8, 36, dict(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)),
    "cfg1_tiny_mlm_bs2": ("tiny_2_2_1.json", 2, 7, 16, 1, 8, dict(masked_language=True)),
    # BASELINE configs[3]: train.py --ranking --shuffle_visual_features, 4 beams + 2 negatives, 7 steps x 36 regions, bs=16/GPU
    # BASELINE configs[4]: bf16 MFMA path + fused AdamW, long-trajectory stress (16 frames x 36 regions), bs=32/GPU; run with --precision bf16
    "cfg5_long_traj_bs32": ("bert_base_6_layer_6_connect.json", 32, 7, 80, 16, 36, dict(ranking=True, traj_judge=True, masked_vision=True, masked_language=True)),
    # inference re-ranking (test.py:144-192): 30 candidate beams of one instruction, T=60, 8 viewpoints x 101 regions, forward only
    "infer_rerank_beam30": ("bert_base_6_layer_6_connect.json", 1, 30, 60, 8, 101, dict(ranking=True, pretrain=False)),
    "cfg4_finetune_rank_bs16": ("bert_base_6_layer_6_connect.json", 16, 6, 80, 7, 36, dict(ranking=True, pretrain=False)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2_full_pretrain_bs8", choices=sorted(WORKLOADS))
    ap.add_argument("--bs", type=int, default=None, help="override items per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra (non-headline) fp32x3 / bf16 measurements of the default run")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="auto|on: replay the step from hipGraphs (1 GPU: one graph; N GPUs: two graphs around the RCCL all-reduce), falling back to eager launches if capture fails; off: eager launches")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32x3"], default="fp32",
                    help="fp32 (default, the headline: the reference's arithmetic on the fp32 matrix instruction); bf16: dense projections "
                         "on bf16 MFMA operands with fp32 accumulation, everything else fp32 (BASELINE configs[4]); fp32x3: fp32 operands, "
                         "each value split exactly into three bf16 terms in registers, six bf16 MFMAs per product (fp32-level error)")
    ap.add_argument("--two-stream", choices=["on", "off"], default="on",
                    help="on: the text side of the model is enqueued on a second HIP stream (ytvln.ops.set_two_stream; results are bit-identical, "
                         "the two sides become two branches of the captured graph and fill each other's idle CUs)")
    ap.add_argument("--h2d", choices=["off", "serial", "overlap", "compact"], default="off",
                    help="also move the batch from pinned host memory to HBM every step (NOT the headline: `value` is quoted with inputs "
                         "resident in HBM): serial = on the compute stream before the step, overlap = on a copy stream under the previous step, "
                         "compact = ship every distinct frame once + un-masked tokens and assemble the batch in HBM (ytvln.batch)")
    ap.add_argument("--loss-aware-heads", action="store_true",
                    help="(next-row experiment, not the headline) decode only rows that carry a masked-token / masked-region "
                         "target; same losses and gradients, fewer FLOPs than the reference's full decode")
    ap.add_argument("--kernel-table", action="store_true", help="print per-shape GEMM timing to stderr")
    ap.add_argument("--eval-dropout-off", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--host-probe", type=int, default=3, metavar="N",
                    help="N extra steps after the timed region (0: none), each enqueued into EMPTY queues (device synchronised first): wall and "
                         "CPU time of the enqueue alone = the host work of a step without back-pressure waits (tools/host8.py)")
    ap.add_argument("--host-wait", choices=["blocking", "spin", "lean"], default="blocking",
                    help="blocking (default): hipDeviceScheduleBlockingSync + the host paced two steps ahead of the device on blocking events "
                         "(ytvln.misc.set_host_wait / StepPacer): a rank sleeps while its GPU works -- <= one core per rank; spin: the runtime's "
                         "default (the host fills the stream's queue and spins for room: two cores per rank in round 5); lean: blocking + "
                         "AMD_DIRECT_DISPATCH=0 (the HIP runtime's own signal thread, which spins a core whenever the device is busy, is replaced by its "
                         "queue thread: 14 ms of CPU per 110 ms step instead of 113, at -4.5 %% throughput on one GPU -- for hosts with < 2 cores per rank)")
    ap.add_argument("--pace", default="event:2", metavar="MODE:DEPTH",
                    help="host pacing under --host-wait blocking (ytvln.misc.StepPacer): MODE event (blocking hipEventSynchronize) or poll (query + sleep "
                         "1 ms), DEPTH = steps the host may run ahead of the device")
    ap.add_argument("--graphs", type=int, default=1, choices=[1, 2],
                    help="1 GPU: number of instantiated copies of the captured step replayed in turn (2: a replay never has to wait for the previous "
                         "launch of the SAME executable graph to finish)")
    ap.add_argument("--loop", choices=["graph", "reference"], default="graph",
                    help="reference (not the headline): the body of the reference's train_epoch as it stands (utils/utils_init.py:199-268) on the "
                         "drop-in modules -- eager launches, model.zero_grad(), every logged scalar read back with float() each step: what an "
                         "import swap alone gives before the loop adopts train_step / hipGraph replay")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="(diagnostic, not a measurement configuration) run the N > 1 code path -- DataParallel wrapper, two-graph step, RCCL "
                         "all-reduce of the whole gradient arena through the C ABI communicator -- in a ONE-rank world on a single GPU")
    return ap.parse_args()


def make_args(flags):
    a = dict(model_name="vilbert", ranking=False, traj_judge=False, masked_vision=False, masked_language=False, pretrain=True,
             num_negatives=2, traj_loss_scale=1.0, not_traj_judge_data=False, local_rank=-1, skip_all_reduce=True,
             weight_decay=0.01, learning_rate=4e-5, no_scheduler=False, ConstantLR=False, gradient_accumulation_steps=1,
             num_epochs=1, warmup_proportion=0.2, cooldown_factor=2.0, resume=False)
    a.update(flags)
    return types.SimpleNamespace(**a)


import threading


class PowerSampler:
    """Package power and shader clock of the device this rank computes on, read from the hwmon node under its PCI address
    (/sys/bus/pci/devices/<bdf>/hwmon/*: power1_input in microwatts, freq1_input in Hz, power1_cap) at 20 Hz by a host thread while the timed
    steps run.  Says on which side of the power limit a run was taken (the fp32 GEMMs lose 15 % on boxes that hold the chip below ~1.2 kW and
    nothing on boxes that do not; LABNOTES 5b).  None when the container does not expose the node."""

    def __init__(self, dev):
        import glob
        import threading
        self.hw = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            nodes = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            self.hw = nodes[0] if nodes else None
        except Exception:
            self.hw = None
        self.p, self.f, self.on = [], [], False
        self.tid, self.cpu_s = None, 0.0
        self._thread = threading.Thread(target=self._run, daemon=True) if self.hw else None

    def _read(self, name):
        try:
            with open(os.path.join(self.hw, name)) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        # (a hwmon read of the SMU busy-waits in the driver for tens of ms: this thread shows up as a core's worth of CPU in the process totals.  It is
        #  the MEASUREMENT's thread, not the path's -- named, and its CPU time is reported separately and excluded from host_cpu_process_ms_per_step)
        try:
            import ctypes
            ctypes.CDLL(None).prctl(15, b"bench-power", 0, 0, 0)
            self.tid = threading.get_native_id()
        except Exception:
            pass
        c0 = time.thread_time()
        try:
            self._loop()
        finally:
            self.cpu_s = time.thread_time() - c0

    def _loop(self):
        while self.on:
            p, f = self._read("power1_input"), self._read("freq1_input")
            if p is not None:
                self.p.append(p / 1e6)
            if f is not None:
                self.f.append(f / 1e6)
            time.sleep(0.05)

    def start(self):
        if self._thread is not None:
            self.on = True
            self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self.on = False
        self._thread.join()
        cap = self._read("power1_cap")
        out = {"source": self.hw + "/power1_input, freq1_input (20 Hz over the timed steps)", "samples": len(self.p)}
        if self.p:
            out.update(mean_w=round(sum(self.p) / len(self.p), 1), max_w=round(max(self.p), 1))
        if cap is not None:
            out["cap_w"] = round(cap / 1e6, 1)
        if self.f:
            out.update(sclk_mhz_mean=round(sum(self.f) / len(self.f), 1), sclk_mhz_min=round(min(self.f), 1))
        return out


class GemmTimer:
    """Brackets every GEMM launch with HIP events on the launch stream (torch's current stream = the stream the C ABI is
    given) and sums algorithmic FLOPs; read out after the timed region."""

    def __init__(self):
        self.records = []
        self.on = False

    def install(self):
        from ytvln import ops
        inner = ops._gemm
        timer = self

        def timed(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw):
            if not timer.on:
                return inner(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            done = inner(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw)
            e1.record()
            timer.records.append((e0, e1, M, N, K, int(transA), int(transB)))
            return done

        ops._gemm = timed
        inner_b = ops._gemm_bf16          # the bf16-resident path's projections (ytvln_gemm_bf16)

        def timed_b(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw):
            if not timer.on:
                return inner_b(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            done = inner_b(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, **kw)
            e1.record()
            timer.records.append((e0, e1, M, N, K, int(transA), int(transB)))
            return done

        ops._gemm_bf16 = timed_b

    def summary(self):
        tot_ms, tot_flop, shapes = 0.0, 0.0, {}
        for e0, e1, M, N, K, ta, tb in self.records:
            ms = e0.elapsed_time(e1)
            fl = 2.0 * M * N * K
            tot_ms += ms
            tot_flop += fl
            s = shapes.setdefault((M, N, K, ta, tb), [0, 0.0, 0.0])
            s[0] += 1; s[1] += ms; s[2] += fl
        return tot_ms, tot_flop, len(self.records), shapes


_THREAD_BIRTH = {}


def mark_threads(stage: str) -> None:
    """Remember during which stage of the run each thread of this process first existed (diagnostic for host_busiest_threads_ms_per_step)."""
    try:
        for tid in os.listdir("/proc/self/task"):
            _THREAD_BIRTH.setdefault(int(tid), stage)
    except OSError:
        pass


def thread_cpu_times() -> dict:
    """{tid: (name, user + system CPU seconds)} of every thread of this process (/proc/self/task): WHICH thread burns the host cores."""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open(f"/proc/self/task/{tid}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tick)
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


def effective_cores() -> int:
    """Host cores this process may really use: min(affinity mask, cgroup CPU quota).  The GPU boxes expose 256 logical CPUs
    behind a 16-CPU cgroup quota; sizing the thread pool by os.cpu_count() there throttles everything to a crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


class FamilyTimer:
    """Brackets the attention / LayerNorm / AdamW entry points of the C ABI with HIP events (eager pass only) and keeps the algorithmic work of
    each call, so that the bench line carries one roofline entry per kernel family (attention by USEFUL matrix FLOPs -- no recompute, no
    padding --, LayerNorm and AdamW by algorithmic bytes against the 8 TB/s HBM figure).  GEMM launches have their own timer (GemmTimer)."""
    FAMILIES = (("ytvln_attn_", "attention"), ("ytvln_ln_", "layernorm"), ("ytvln_adamw", "adamw"))

    def __init__(self):
        self.records = []
        self.on = False

    def install(self):
        from ytvln import ops
        inner = ops.call
        timer = self

        def timed(name, *args):
            fam = None
            if timer.on:
                for prefix, f in timer.FAMILIES:
                    if name.startswith(prefix):
                        fam = f
                        break
            if fam is None:
                return inner(name, *args)
            work = 0.0
            if fam == "layernorm":           # (.., rows, H, ..): 12 B read/written + 4 B saved per element forward, 20 B backward
                if name == "ytvln_ln_bwd_bf16":          # bf16 rows: dy, s in; ds (and dx under dropout) out -> 8 B per element
                    rows, H = int(args[9]), int(args[10])
                    work = rows * H * 8.0
                elif name.endswith("_bf16"):             # x, res in; y, s out -> 8 B per element
                    rows, H = int(args[8]), int(args[9])
                    work = rows * H * 8.0
                else:
                    rows, H = int(args[8]), int(args[9])
                    work = rows * H * (20.0 if "bwd" in name else 16.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = inner(name, *args)
            e1.record()
            timer.records.append((fam, e0, e1, work))
            return r

        ops.call = timed

    def summary(self):
        out = {}
        for fam, e0, e1, work in self.records:
            d = out.setdefault(fam, [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1)
            d[1] += work
            d[2] += 1
        return out


def attention_useful_flops(cfg, N, T, R):
    """Useful matrix FLOPs of all attention sites of one training step: 4 Tq Tk (heads x d) per (pair) forward + 8 backward (dV, dP, dQ, dK),
    i.e. without the backward's score recompute and without tile padding (vilbert/vilbert.py:276-307, 413-440)."""
    f = 12.0 * N
    text = cfg.num_hidden_layers * T * T * cfg.hidden_size
    image = cfg.v_num_hidden_layers * R * R * cfg.v_hidden_size
    co = len(cfg.v_biattention_id) * 2 * T * R * cfg.bi_hidden_size
    return f * (text + image + co)


def cpu_baseline(workload, budget_s: float = 45.0):
    """Oracle (port of the reference's PyTorch CPU path) on the host cores: bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vilbert_ref as O
    from ytvln import synth
    cfgname, _, K, T, frames, boxes, flags = WORKLOADS[workload]
    cores = effective_cores()
    torch.set_num_threads(cores)
    cfgd = json.load(open(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname)))
    ocfg = O.RefConfig(**cfgd)
    oflags = O.TaskFlags(**flags)
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")))["Lily/" + cfgname]["shapes"]
    W = synth.make_weights({k: tuple(v) for k, v in shapes.items()}, seed=1)
    S = {k: torch.from_numpy(v).clone() for k, v in W.items()}
    bs = 1 if "full" in workload else 2
    batch = synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=1234, ignore_rank_frac=0.0))
    st = O.AdamWState()
    times = []
    begin = time.perf_counter()
    for i in range(3):
        t0 = time.perf_counter()
        O.train_step(S, ocfg, oflags, batch, st, 4e-5, drop=True)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - begin + times[-1] > budget_s:      # bounded sample: never let the checker dominate the run
            break
    med = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"value": round(bs * K / med, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle/vilbert_ref.py (port of the reference's PyTorch CPU path), {cfgname}, bs={bs} x K={K} = {bs * K} pairs, "
                      f"T={T} R={frames * boxes}, fp32, dropout on, fwd + losses + bwd + AdamW; median of {max(1, len(times) - 1)} step(s) "
                      f"after {1 if len(times) > 1 else 0} warm-up ({med:.2f} s/step)"}


class stdout_to_stderr:
    """File-descriptor level: native libraries (gloo's connection notice, RCCL's version banner under NCCL_DEBUG=INFO) print to fd 1 while
    they initialise; the bench's stdout carries exactly ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def self_launch(n: int):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks of ONE node under torch.distributed.run -- the command the
    reference's README uses for its own multi-GPU runs (README.md:98-100, `python -m torch.distributed.launch --nproc_per_node ...`)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


_REAL_STDOUT = None


def emit(line: str):
    """The ONE JSON line, on the process's original stdout (see main: fd 1 itself points at stderr while the bench runs)."""
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def main():
    a = parse()
    if a.host_wait == "lean" and os.environ.get("AMD_DIRECT_DISPATCH") != "0":
        os.environ["AMD_DIRECT_DISPATCH"] = "0"          # read when the HIP runtime initialises: start over with it set
        os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], os.environ)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)                        # does not return
    # Native libraries print to fd 1 whenever they like (gloo's connection notice at start-up, RCCL's version banner under
    # NCCL_DEBUG=INFO when the communicator is torn down): keep the original stdout for the JSON line and point fd 1 at stderr.
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without a launcher)")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # host threads: the per-process share of the cores this cgroup may really use (N ranks on one node share them)
    torch.set_num_threads(max(1, effective_cores() // max(1, world)))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # (ranks may share a device only in the gloo self-test below)
    mark_threads("interpreter start-up (python, torch import)")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    torch.zeros(1, device=dev)
    mark_threads("HIP runtime initialisation (first device allocation)")
    from ytvln import misc as yt_misc
    host_wait = a.host_wait
    if a.host_wait in ("blocking", "lean"):
        try:
            yt_misc.set_host_wait(True, dev_index)
        except RuntimeError as e:          # the flag is an optimisation of the HOST side; never a reason to lose the measurement
            print(f"[bench] rank {rank}: blocking host wait unavailable ({e}); spinning", file=sys.stderr)
            host_wait = "spin (blocking unavailable)"
    from ytvln import distributed as D
    collective = D.default_collective()
    dp_wrap = world > 1 or a.dp_selftest
    rccl_log = None
    if dp_wrap and os.environ.get("YTVLN_BENCH_RCCL_INFO", "1") != "0" and "NCCL_DEBUG" not in os.environ:
        # the N > 1 line explains itself: RCCL's own choice of algorithm / protocol for the gradient all-reduce, read back from its log
        # (a FILE: the bench's stdout carries exactly one JSON line)
        rccl_log = f"/tmp/ytvln_rccl_{os.getpid()}.log"
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,TUNING,GRAPH", NCCL_DEBUG_FILE=rccl_log)
    if dp_wrap:
        # Data plane: the C ABI's own RCCL communicator (ytvln_rccl_*), torch.distributed = env:// rendezvous + gloo control plane.
        # YTVLN_DP_COLLECTIVE=torch runs the exchange through torch.distributed instead ("nccl" = RCCL on ROCm).
        # YTVLN_DIST_BACKEND=gloo with more ranks than GPUs exists only to exercise the N > 1 code path on ONE GPU (RCCL refuses
        # duplicate devices); it is never a measurement configuration.
        if world > ndev:
            if os.environ.get("YTVLN_DIST_BACKEND") != "gloo":
                raise SystemExit(f"{world} ranks need {world} GPUs (found {ndev})")
            collective = "torch"
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        with stdout_to_stderr():
            D.init_distributed(backend="nccl" if (collective == "torch" and world <= ndev) else "gloo", force=True)

    from ytvln import ops as yt_ops
    from ytvln import synth, utils_init
    from ytvln.distributed import DataParallel
    yt_ops.set_matmul_precision(a.precision)
    yt_ops.set_two_stream(a.two_stream == "on")
    from ytvln.lily import Lily
    from ytvln.vilbert import BertConfig
    from ytvln.vilbert_init import get_optimization

    cfgname, bs, K, T, frames, boxes, flags = WORKLOADS[a.workload]
    bs = a.bs or bs
    args = make_args(flags)
    args.local_rank = local_rank if world > 1 else -1
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname))
    cfg.args = args
    torch.manual_seed(1234)                      # identical initial weights on every rank (DDP would broadcast rank 0's)
    model = Lily(cfg).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    model.train()
    if a.eval_dropout_off:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    batch = synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=1234 + rank,
                                            finetune_heading=not args.pretrain), dev)
    runner = model
    if dp_wrap:
        with stdout_to_stderr():
            runner = DataParallel(model, broadcast=True, collective=collective, always_exchange=a.dp_selftest)
            torch.cuda.synchronize()
    opt, sched, _, _ = get_optimization(args, model, a.steps + a.warmup + 1, None)
    if dp_wrap:
        runner.attach(opt)

    def control_reduce(x: float, op) -> float:
        """max / min of a host scalar over ranks on the control plane (CPU tensor: gloo; device tensor: nccl)."""
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    mark_threads("model / optimizer construction")
    timer = GemmTimer()
    ftimer = FamilyTimer()
    if not a.no_kernel_timing:
        timer.install()
        ftimer.install()

    infer = a.workload.startswith("infer")
    if infer:
        model.eval()
        inputs = utils_init.get_model_input(batch, True)

        def step(i):      # forward only (test.py:144-166): scores of the candidate paths stay on the device
            with torch.no_grad():
                return runner(*inputs)["ranking"].sum(), None
    else:
        def step(i):
            return utils_init.train_step(runner, opt, sched, batch, args, i, all_options=True, loss_aware_heads=a.loss_aware_heads)

    if a.loop == "reference" and not infer:
        from ytvln.utils_init import compute_metrics_independent, get_model_input

        def step(i):      # noqa: F811 -- utils/utils_init.py:199-268 line by line (tensorboard / logger calls = the float() reads they imply)
            outputs = runner(*get_model_input(batch))
            loss = torch.tensor(0, device=dev).float()
            rm = {"loss": {}, "accuracy": {}}
            if args.masked_vision:
                loss += compute_metrics_independent(batch, outputs, "vision", args, None, rm)
            if args.masked_language:
                loss += compute_metrics_independent(batch, outputs, "language", args, None, rm)
            if args.ranking:
                loss += compute_metrics_independent(batch, outputs, "ranking", args, None, rm)
            if args.traj_judge:
                loss += args.traj_loss_scale * compute_metrics_independent(batch, outputs, "traj", args, None, rm)
            rm["loss/train"] = torch.tensor(0, device=dev).detach().float()
            for item in rm["loss"].values():
                rm["loss/train"] += item
            loss.backward()
            opt.step()
            sched.step()
            runner.zero_grad()
            logged = [float(sched.get_last_lr()[0]), float(rm["loss/train"])] + [float(v) for v in rm["accuracy"].values()] + \
                     [float(v) for v in rm["loss"].values()]          # writer.add_scalar(...) + logger.info(f"{...:.2f}") of the reference
            return loss.detach(), logged

    use_graph = a.graph in ("on", "auto") and not infer and a.loop == "graph"
    execution = "eager launches" if a.loop == "graph" else \
        "the reference's train_epoch body as it stands: eager launches, model.zero_grad(), logged scalars read back every step"
    eager_step = step
    if use_graph:
        # world == 1: hipGraph replay of the whole step (forward, losses, backward, fused AdamW); the host only uploads the
        # LR-dependent hyper-parameters and advances the schedule.  world > 1: two graphs (forward+backward | AdamW) with the
        # RCCL gradient all-reduce between them, outside any graph (ytvln.distributed.GraphedTrainStep).  Dropout masks still
        # change every replay (device-side counter).  Capture is an optimisation, never a requirement: any rank that cannot
        # capture sends every rank back to eager launches (the collective sizes differ between the two modes).
        ok = True
        try:
            for i in range(2):                                 # eager steps: build the optimizer arenas, warm the allocator
                eager_step(i)
            torch.cuda.synchronize()
            if not dp_wrap:
                graphs = []
                for _ in range(a.graphs):
                    graph = torch.cuda.CUDAGraph()
                    static = {}
                    with torch.cuda.graph(graph):
                        static["loss"], _ = utils_init.train_step(runner, opt, None, batch, args, 0, all_options=True,
                                                                     loss_aware_heads=a.loss_aware_heads)
                    torch.cuda.synchronize()
                    graphs.append((graph, static))

                def graph_step(i):
                    graph, static = graphs[i % len(graphs)]
                    opt.prepare_replay()
                    graph.replay()
                    sched.step()
                    return static["loss"], None
            else:
                from ytvln.distributed import GraphedTrainStep
                gs = GraphedTrainStep(runner, opt, lambda backward=None: utils_init.train_step(
                    runner, opt, None, batch, args, 0, all_options=True, loss_aware_heads=a.loss_aware_heads, optimizer_step=False,
                    backward=backward)[0])

                def graph_step(i):
                    return gs.step(sched), None
        except Exception as e:
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            ok = False
        ok = control_reduce(1.0 if ok else 0.0, dist.ReduceOp.MIN) > 0.5
        if ok:
            step = graph_step
            how = (f"{len(gs.graphs)} hipGraphs per step (forward + backward in {len(gs.graphs)} phases), every phase's gradients all-reduced and its "
                   "parameters updated (fused AdamW per group) "
                   + ("on a communication stream under the next phases" if runner.comm is not None else "after its graph, in stream order")) \
                if (dp_wrap and gs.mode == "phased") else \
                "two hipGraphs per step (forward+backward | AdamW) with the RCCL all-reduce between them"
            execution = "hipGraph replay of the captured step" if not dp_wrap else \
                (f"{how} [{gs.mode}; exchange: {'ytvln_rccl_* C ABI' if runner.comm is not None else 'torch.distributed ' + dist.get_backend()}]")
        else:
            torch.cuda.synchronize()
            opt.zero_grad()
            use_graph = False
            step = eager_step

    mark_threads("eager steps + graph capture")
    for i in range(a.warmup):
        loss, _ = step(i)
    torch.cuda.synchronize()
    mark_threads("warm-up steps (graph replays)")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = ftimer.on = not use_graph             # graph replays cannot bracket single kernels: see the eager pass below
    if a.h2d != "off":
        # PCIe-inclusive variant: the step consumes `batch` (static device tensors); a pinned host copy is re-uploaded every step.
        host = [t.cpu().pin_memory() if torch.is_tensor(t) else t for t in batch]
        h2d_bytes = sum(t.numel() * t.element_size() for t in host if torch.is_tensor(t))
        copy_stream = torch.cuda.Stream()
        inner_step = step
        if a.h2d == "compact":
            from ytvln import batch as yt_batch
            pool = [torch.from_numpy(np.ascontiguousarray(x)).pin_memory() for x in
                    synth.make_pool(bs, K, T, frames, boxes, seed=4321 + rank)]
            h2d_bytes = sum(t.numel() * t.element_size() for t in pool)

            def step(i):   # noqa: F811
                pf, pb, pp, pm, index, tok, tmask = [t.to(dev, non_blocking=True) for t in pool]
                f, bx, pr, m = yt_batch.expand_options(pf, pb, pp, pm, index)
                b = list(batch)
                b[1], b[2], b[3], b[4], b[6], b[7] = f, bx, m, pr, tok, tmask.bool()
                b = yt_batch.mask_batch(b)
                for k in (1, 2, 3, 4, 5, 6, 7, 8):                 # into the static tensors the (graphed) step reads
                    batch[k].copy_(b[k].view(batch[k].shape))
                return inner_step(i)
        elif a.h2d == "serial":
            def step(i):   # noqa: F811
                for d, h in zip(batch, host):
                    if torch.is_tensor(d):
                        d.copy_(h, non_blocking=True)
                return inner_step(i)
        else:
            staged = [torch.empty_like(t) if torch.is_tensor(t) else t for t in batch]
            ready = torch.cuda.Event()

            def upload():
                with torch.cuda.stream(copy_stream):
                    for d, h in zip(staged, host):
                        if torch.is_tensor(d):
                            d.copy_(h, non_blocking=True)
                    ready.record(copy_stream)
            upload()

            def step(i):   # noqa: F811
                torch.cuda.current_stream().wait_event(ready)          # the staged batch for this step has landed
                for d, st in zip(batch, staged):                       # device-to-device swap into the tensors the step reads
                    if torch.is_tensor(d):
                        d.copy_(st, non_blocking=True)
                done = torch.cuda.Event(); done.record()
                copy_stream.wait_event(done)
                upload()                                               # next batch travels under this step's compute
                return inner_step(i)
    if a.h2d != "off":
        for i in range(2):                         # the upload / assembly buffers are new: let the allocator settle before timing
            step(a.warmup + i)
        torch.cuda.synchronize()
    power = PowerSampler(dev) if rank == 0 else None
    if power is not None:
        power.start()
    pace_mode, pace_depth = a.pace.split(":")
    pacer = yt_misc.StepPacer(int(pace_depth), pace_mode) if a.host_wait in ("blocking", "lean") else None
    t_step = t_tick = 0.0
    threads0 = thread_cpu_times()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    pc0 = time.process_time()
    enq = []
    for i in range(a.steps):
        e0 = time.perf_counter()
        c_0 = time.thread_time()
        loss, _ = step(a.warmup + i)
        c_1 = time.thread_time()
        if pacer is not None:
            pacer.tick()          # sleep until step i - depth has finished: the queue never fills, nothing spins
        t_step += c_1 - c_0
        t_tick += time.thread_time() - c_1
        enq.append(time.perf_counter() - e0)
    host_enqueue = time.perf_counter() - t0          # wall time until the last step is enqueued: INCLUDES waiting for room in the stream's queue
    host_cpu = time.thread_time() - c0               # CPU time this thread spent enqueueing: what N ranks on one host really compete for
    if pacer is not None:
        pacer.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    power_info = power.stop() if power is not None else None
    host_cpu_process = time.process_time() - pc0     # every thread of this rank until the steps have drained: + RCCL proxy / HIP runtime threads
    sampler_cpu = power.cpu_s if power is not None else 0.0
    host_cpu_process = max(0.0, host_cpu_process - sampler_cpu)          # minus the bench's own power-sampling thread (see PowerSampler._run)
    threads1 = thread_cpu_times()
    mark_threads("timed steps")
    host_threads = sorted(((round(1000.0 * (threads1[t][1] - threads0.get(t, (threads1[t][0], 0.0))[1]) / a.steps, 2),
                            threads1[t][0] + (" [main]" if t == os.getpid() else "") + " <" + _THREAD_BIRTH.get(t, "?") + ">") for t in threads1),
                          reverse=True)[:4]          # the four busiest threads of this rank during the timed region: ms of CPU per step, name
    timer.on = ftimer.on = False
    elapsed = control_reduce(elapsed, dist.ReduceOp.MAX)
    host_probe = None
    if a.host_probe > 0:
        pw, pc, pd = [], [], []
        for i in range(a.host_probe):
            torch.cuda.synchronize()
            w0, k0 = time.perf_counter(), time.thread_time()
            loss, _ = step(a.warmup + a.steps + i)
            w1 = time.perf_counter()
            pc.append(1000.0 * (time.thread_time() - k0))
            torch.cuda.synchronize()
            pw.append(1000.0 * (w1 - w0))
            pd.append(1000.0 * (time.perf_counter() - w1))
        host_probe = {"steps": a.host_probe, "enqueue_wall_ms": {"p50": round(float(np.percentile(pw, 50)), 2), "max": round(max(pw), 2)},
                      "enqueue_cpu_ms": {"p50": round(float(np.percentile(pc, 50)), 2), "max": round(max(pc), 2)},
                      "drain_ms": {"p50": round(float(np.percentile(pd, 50)), 2), "max": round(max(pd), 2)},
                      "note": "each step enqueued into empty queues: enqueue_* = until the step() call returns (host work of one step, no back-pressure "
                              "waits); drain_ms = from there until the device has finished that step (the whole step was still ahead)"}
    final_loss = float(loss)
    assert np.isfinite(final_loss), "training diverged"
    # data parallel: after the timed steps every replica must hold the same parameters (each rank saw different data, so this holds only
    # if every gradient was exchanged): max - min over ranks of a checksum of all parameters, 0.0 when the replicas are bit-identical
    replica_spread = None
    if dp_wrap and not infer:
        with torch.no_grad():
            chk = float(sum(p.detach().double().sum() for p in model.parameters()))
        replica_spread = control_reduce(chk, dist.ReduceOp.MAX) - control_reduce(chk, dist.ReduceOp.MIN)
    dp_diag = None
    if dp_wrap and not infer:
        dp_diag = {}
        try:
            flat = opt.flat_grad()
            nbytes = flat.numel() * flat.element_size()
            # (a) the exchange alone: the whole gradient arena, as the step issues it, 3 repetitions between events on the current stream
            reps = 3
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cap = max(1, (256 << 20) // flat.element_size())
            slices = [(lo, min(lo + cap, flat.numel())) for lo in range(0, flat.numel(), cap)]
            e0.record()
            for _ in range(reps):
                if runner.comm is not None:
                    runner.comm.all_reduce_slices(flat, slices)
                else:
                    for lo, hi in slices:
                        dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = control_reduce(e0.elapsed_time(e1) / reps, dist.ReduceOp.MAX)
            flat.zero_()
            dp_diag["allreduce_bytes"] = nbytes
            dp_diag["allreduce_alone_ms"] = round(ar_ms, 3)
            dp_diag["allreduce_alg_gbps"] = round(nbytes / (ar_ms * 1e-3) / 1e9, 1)
            dp_diag["allreduce_bus_gbps"] = round(nbytes / (ar_ms * 1e-3) / 1e9 * (2.0 * (world - 1) / world if world > 1 else 1.0), 1)
            # (b) how much of it the step exposes: two more (untimed) steps with events around the wait for the communication stream
            if use_graph:
                gs.profile = True
                exp = []
                for i in range(2):
                    step(a.warmup + a.steps + 10 + i)
                    x = gs.exposed_exchange_ms()
                    if x is not None:
                        exp.append(x)
                gs.profile = False
                if exp:
                    dp_diag["exchange_exposed_ms"] = round(control_reduce(sum(exp) / len(exp), dist.ReduceOp.MAX), 3)
                    dp_diag["exchange_form"] = gs.mode
        except Exception as e:       # diagnostics never take the measurement down
            dp_diag["error"] = f"{type(e).__name__}: {e}"
        if rccl_log and os.path.exists(rccl_log):
            import re
            algo = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree"}
            proto = {0: "LL", 1: "LL128", 2: "Simple"}
            seen, chans, transports = [], None, set()
            for line in open(rccl_log, errors="replace"):
                m = re.search(r"(\d+) Bytes -> Algo (\d+) proto (\d+)", line)
                if m:
                    t = (int(m.group(1)), algo.get(int(m.group(2)), m.group(2)), proto.get(int(m.group(3)), m.group(3)))
                    if t not in seen:
                        seen.append(t)
                m = re.search(r"(\d+) coll channels", line)
                if m:
                    chans = int(m.group(1))
                m = re.search(r"via (P2P/\w+|SHM\S*|NET/\S+)", line)
                if m:
                    transports.add(m.group(1))
            dp_diag["rccl"] = {"choices": [{"bytes": b, "algo": al, "proto": pr} for b, al, pr in seen[-6:]], "coll_channels": chans,
                               "transports": sorted(transports), "log": rccl_log}
    if infer:
        execution = "eager launches, eval mode, forward only"
    roofline_note = "HIP events around every GEMM launch during the timed steps"
    if use_graph and not a.no_kernel_timing:
        n_prof = min(a.steps, 3)
        yt_ops.set_two_stream(False)          # single kernels are timed alone on the chip (one stream), whatever the timed replays used
        eager_step(a.warmup + a.steps)        # untimed: eager launches allocate outside the graph's memory pool the first time
        torch.cuda.synchronize()
        timer.on = ftimer.on = True
        for i in range(n_prof):
            eager_step(a.warmup + a.steps + 1 + i)
        torch.cuda.synchronize()
        timer.on = ftimer.on = False
        yt_ops.set_two_stream(a.two_stream == "on")
        roofline_note = f"HIP events around every GEMM launch during {n_prof} eager steps run right after the timed graph replays"

    pairs_per_step = bs * K * world
    value = pairs_per_step * a.steps / elapsed
    out = {
        "metric": "pretrain samples/sec (traj-instr pairs)", "value": round(value, 3), "unit": "pairs/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000.0 * elapsed / a.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16 activations / gradients / logits / weight copies in HBM, bf16 MFMA, f32 accumulate / softmax / LayerNorm statistics / loss reductions / master weights / optimizer",
                  "fp32x3": "f32 operands split exactly into 3 bf16 terms in registers, 6 bf16 MFMAs per product, f32 accumulate (projections); "
                            "f32 everywhere else"}[a.precision], "data": "synthetic",
        "config": {"workload": a.workload, **({"gradient_exchange": ("ytvln_rccl_* (C ABI, " + os.path.basename(runner.comm.library) + ")") if runner.comm is not None
                                                  else "torch.distributed " + dist.get_backend(), "dp_selftest": bool(a.dp_selftest)} if dp_wrap else {}), "model_config": cfgname, "params": n_params, "items_per_gpu": bs, "options_per_item": K,
                   "pairs_per_gpu": bs * K, "global_pairs": pairs_per_step, "tokens": T, "regions": frames * boxes, "feature_dim": 2048,
                   "losses": [k for k, v in flags.items() if v], "dropout": not a.eval_dropout_off,
                   "optimizer": "fused AdamW (HF formula) + WarmupLinear", "parallelism": f"dp{world}",
                   "execution": execution + (", text side on a second HIP stream (two graph branches)" if a.two_stream == "on" else ""), "heads": "loss-aware rows (extension)" if a.loss_aware_heads else "all rows (reference)"},
        "items_per_s": round(value / K, 3), "final_loss": round(final_loss, 4),
        "host_enqueue_ms_per_step": round(1000.0 * control_reduce(host_enqueue, dist.ReduceOp.MAX) / a.steps, 2),
        "host_cpu_ms_per_step": round(1000.0 * control_reduce(host_cpu, dist.ReduceOp.MAX) / a.steps, 2),
        "host_cpu_process_ms_per_step": round(1000.0 * host_cpu_process / a.steps, 2),          # this rank, all threads (incl. RCCL proxy / runtime)
        "host_enqueue_ms_tail": {"p50": round(1000.0 * float(np.percentile(enq, 50)), 2), "max": round(1000.0 * max(enq), 2)},   # per step, this rank
        "host_cores_available": effective_cores(),
        "host_wait": host_wait + (f", host paced {a.pace} (StepPacer mode:depth)" if pacer is not None else ""),
        "host_thread_cpu_ms_per_step": {"in_step_call": round(1000.0 * t_step / a.steps, 2), "in_pacer": round(1000.0 * t_tick / a.steps, 2)},
        "host_busiest_threads_ms_per_step": [{"thread": n, "cpu_ms": v} for v, n in host_threads],
        "bench_power_sampler_cpu_ms_per_step": round(1000.0 * sampler_cpu / a.steps, 2),          # excluded from host_cpu_process_ms_per_step
        "hbm_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1),
        "power": power_info,
    }
    if host_probe is not None:
        out["host_probe"] = host_probe
    if replica_spread is not None:
        out["config"]["replica_checksum_spread"] = replica_spread
    if dp_diag is not None:
        out["data_parallel"] = dp_diag
    if dp_wrap and use_graph and getattr(gs, "mode", "") == "phased":
        # MB of gradients per exchange group, in the order they go out (the last one is the exposed tail)
        out["config"]["exchange_groups_mb"] = [round(4e-6 * sum(hi - lo for lo, hi in g), 1) for g in gs._group_slices]
    if a.h2d != "off":
        out["h2d"] = {"mode": a.h2d, "bytes_per_step": h2d_bytes, "note": "value INCLUDES the host->HBM upload of the batch; not the headline"}
    if "full" in a.workload and T == 80 and frames * boxes == 288 and not a.loss_aware_heads and a.precision in ("fp32", "fp32x3"):   # (FLOP count is the full-decode one)
        out["model_tflops"] = round(value * TRAIN_GFLOP_PER_PAIR / 1000.0, 2)
        if a.precision == "fp32":
            out["model_mfma_frac"] = round(value / world * TRAIN_GFLOP_PER_PAIR / 1000.0 / PEAK_F32_MFMA_TFLOPS, 4)
    if not a.no_kernel_timing and timer.records:
        ms, flop, n, shapes = timer.summary()
        ach = flop / (ms * 1e-3) / 1e12
        traffic, traffic_note = None, None
        try:     # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (cannot be collected in-process): the newest committed summary
            import glob
            import re
            # fp32: roundN_pmc_summary.json (cfg 2); bf16: roundN_bf16_pmc_summary.json (collected at cfg 5: only quoted for that workload)
            pat = r"round\d+_pmc_summary\.json" if a.precision == "fp32" else r"round\d+_bf16_pmc_summary\.json"
            cands = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_summary.json")) if re.fullmatch(pat, os.path.basename(f))),
                           key=lambda f: int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
            pmc = json.load(open(cands[-1]))
            if a.precision == "fp32" or (a.precision == "bf16" and a.workload.startswith("cfg5")):
                traffic = pmc["kernels"]["gemm_dma" if a.precision == "fp32" else "gemm_bf16"]["hbm_side_bytes_per_launch"]
                traffic_note = f"profiles/{os.path.basename(cands[-1])}: (2 x FETCH_SIZE + WRITE_SIZE) per fast-path GEMM launch, separate --pmc passes"
        except (OSError, KeyError, ValueError, IndexError, AttributeError):
            pass
        # fp32x3: six bf16 matrix instructions per algorithmic product -> the ceiling for algorithmic FLOPs is the bf16 peak / 6
        peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "fp32x3": round(PEAK_BF16_MFMA_TFLOPS / 6.0, 1)}[a.precision]
        kname = {"fp32": "ytvln::gemm_dma_kernel (v_mfma_f32_32x32x2_f32)",
                 "bf16": "ytvln::gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16; bf16 operands read in place, transposed operands by ds_read_b64_tr_b16)",
                 "fp32x3": "ytvln::gemm_dma_kernel<X3> (6 x v_mfma_f32_32x32x16_bf16 per product; peak = bf16 dense peak / 6)"}[a.precision]
        if a.precision == "fp32x3":
            traffic, traffic_note = None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "round1_n_fp32x3_pmc_summary.json")))
                traffic = pmc["kernels"]["gemm_dma"]["hbm_side_bytes_per_launch"]
                traffic_note = "profiles/round1_n_fp32x3_pmc_summary.json: (2 x FETCH_SIZE + WRITE_SIZE) per fast-path GEMM launch, separate --pmc passes"
            except (OSError, KeyError, ValueError):
                pass
        out["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2),
                           "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                           "traffic": traffic, "traffic_static": True, "traffic_source": traffic_note, "launches": n, "avg_launch_us": round(1000.0 * ms / n, 2),
                           "avg_launch_gflop": round(flop / n / 1e9, 3), "measured": roofline_note}
        # one entry per kernel family, same eager pass: enough to recompute every fraction from this line alone
        nsteps_prof = max(1, (n_prof if use_graph else a.steps))
        fams = {"gemm": {"bound": "mfma", "ms_per_step": round(ms / nsteps_prof, 3), "launches_per_step": round(n / nsteps_prof, 1),
                         "work_per_step": round(flop / nsteps_prof / 1e12, 4), "work_unit": "TFLOP (2MNK)", "achieved": round(ach, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4)}}
        fsum = ftimer.summary()
        if "attention" in fsum and not infer:
            fms, _, fn = fsum["attention"]
            useful = attention_useful_flops(cfg, bs * K, T, frames * boxes) * nsteps_prof
            apeak = PEAK_BF16_MFMA_TFLOPS if a.precision == "bf16" else PEAK_F32_MFMA_TFLOPS
            fams["attention"] = {"bound": "mfma", "ms_per_step": round(fms / nsteps_prof, 3), "launches_per_step": round(fn / nsteps_prof, 1),
                                 "work_per_step": round(useful / nsteps_prof / 1e12, 4), "work_unit": "TFLOP useful (12 Tq Tk hd per pair and site: no recompute, no padding)",
                                 "achieved": round(useful / (fms * 1e-3) / 1e12, 2), "peak": apeak, "unit": "TFLOP/s",
                                 "frac": round(useful / (fms * 1e-3) / 1e12 / apeak, 4)}
        if "layernorm" in fsum:
            fms, fbytes, fn = fsum["layernorm"]
            fams["layernorm"] = {"bound": "hbm", "ms_per_step": round(fms / nsteps_prof, 3), "launches_per_step": round(fn / nsteps_prof, 1),
                                 "work_per_step": round(fbytes / nsteps_prof / 1e9, 3), "work_unit": "GB algorithmic (fp32: 16 B/element forward incl. the saved sum, 20 B backward; bf16 rows: 8 B/element)",
                                 "achieved": round(fbytes / (fms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(fbytes / (fms * 1e-3) / 1e9 / 8000.0, 4)}
        if "adamw" in fsum:
            fms, _, fn = fsum["adamw"]
            fl = opt.flat_grad()
            abytes = 28.0 * (fl.numel() if fl is not None else n_params) * nsteps_prof
            fams["adamw"] = {"bound": "hbm", "ms_per_step": round(fms / nsteps_prof, 3), "launches_per_step": round(fn / nsteps_prof, 1),
                             "work_per_step": round(abytes / nsteps_prof / 1e9, 3), "work_unit": "GB algorithmic (28 B per parameter that has a gradient)",
                             "achieved": round(abytes / (fms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(abytes / (fms * 1e-3) / 1e9 / 8000.0, 4)}
        out["roofline"]["families"] = fams
        if a.kernel_table and rank == 0:
            rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
            print(f"{'M':>7} {'N':>6} {'K':>6} tA tB {'calls':>6} {'ms':>9} {'TF/s':>7}", file=sys.stderr)
            for (M, N, Kk, ta, tb), (c, msx, fl) in rows[:40]:
                print(f"{M:7d} {N:6d} {Kk:6d} {ta:2d} {tb:2d} {c:6d} {msx:9.3f} {fl / msx / 1e9:7.1f}", file=sys.stderr)
    if world == 1 and not dp_wrap and a.precision == "fp32" and use_graph and a.h2d == "off" and not a.no_variants:
        # NOT the headline: the same captured step with the opt-in fp32x3 projections (fp32 operands, three exact bf16 terms per value,
        # six bf16 MFMAs per product; same parity bar as the native instruction, LABNOTES.md 5a), timed after everything above.
        def time_variant(mode, base):
            """pairs/s and ms/step of the same step captured again under another projection arithmetic (after the headline is timed)"""
            yt_ops.set_matmul_precision(mode)
            for i in range(2):
                eager_step(base + i)
            torch.cuda.synchronize()
            gv = torch.cuda.CUDAGraph()
            stv = {}
            with torch.cuda.graph(gv):
                stv["loss"], _ = utils_init.train_step(runner, opt, None, batch, args, 0, all_options=True,
                                                        loss_aware_heads=a.loss_aware_heads)
            torch.cuda.synchronize()

            def stepv():
                opt.prepare_replay()
                gv.replay()
                sched.step()
            for _ in range(2):
                stepv()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(a.steps):
                stepv()
            torch.cuda.synchronize()
            ev = time.perf_counter() - tv
            assert np.isfinite(float(stv["loss"])), f"{mode} variant diverged"
            del gv
            return round(bs * K * a.steps / ev, 3), round(1000.0 * ev / a.steps, 3)

        try:
            v3, ms3 = time_variant("fp32x3", a.warmup + a.steps + 8)
            # accuracy evidence in the same line: one projection-sized product in both arithmetics against the fp64 product
            gq = torch.Generator(device="cpu").manual_seed(7)
            Aq = torch.randn(4096, 1024, generator=gq).to(dev)
            Bq = torch.randn(1024, 1024, generator=gq).to(dev)
            refq = Aq.double() @ Bq.double().t()
            errs = {}
            for mode in ("fp32", "fp32x3"):
                yt_ops.set_matmul_precision(mode)
                Cq = torch.empty(4096, 1024, device=dev)
                yt_ops._gemm(Aq, 1024, 0, Bq, 1024, 1, Cq, 1024, 4096, 1024, 1024)
                errs[mode] = float((Cq.double() - refq).abs().max() / refq.abs().max())
            yt_ops.set_matmul_precision("fp32x3")
            if "roofline" in out:
                out["roofline"]["trace_note"] = ("in a kernel trace of this command the headline kernel is every gemm_dma_kernel<..., false, false> "
                                                 "instantiation; the <..., false, true> launches belong to variants.fp32x3 (--no-variants omits them)")
            out["variants"] = {"fp32x3": {"value": v3, "unit": "pairs/s", "ms_per_step": ms3,
                                          "note": "opt-in --precision fp32x3, not the headline: fp32 operands split exactly into 3 bf16 terms in "
                                                  "registers, 6 bf16 MFMAs per product, f32 accumulate; meets the fp32 parity bar (LABNOTES.md 5a)",
                                          "gemm_4096x1024x1024_max_err_over_max_vs_f64": {"native_f32_mfma": float(f"{errs['fp32']:.3e}"),
                                                                                          "fp32x3": float(f"{errs['fp32x3']:.3e}")}}}
        except Exception as e:      # never let the extra measurement endanger the headline line
            out["variants"] = {"fp32x3": {"error": f"{type(e).__name__}: {e}"}}
        try:        # bf16 operands on the MFMA (the arithmetic of BASELINE configs[4], the reference's --amp analogue) at the headline shape
            vb, msb = time_variant("bf16", a.warmup + a.steps + 16)
            out.setdefault("variants", {})["bf16"] = {"value": vb, "unit": "pairs/s", "ms_per_step": msb,
                                                      "note": "opt-in --precision bf16, not the headline: the bf16-resident path (bf16 activations / gradients / weight copies, "
                                                              "f32 accumulate, softmax, statistics, logits, master weights; parity: the cfg-5 goldens, bf16 bar)"}
        except Exception as e:
            out.setdefault("variants", {})["bf16"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            yt_ops.set_matmul_precision("fp32")
        if a.workload == "cfg2_full_pretrain_bs8":
            # BASELINE configs[4] at ITS size (bs 32 items = 224 pairs per GPU, 16 frames x 36 = 576 regions, bf16 MFMA operands) in its own
            # process after everything above -- so that the driver's default run sees that number too, with its own roofline against the
            # dense bf16 peak.  NOT the headline.
            try:
                import subprocess
                torch.cuda.empty_cache()
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", "cfg5_long_traj_bs32", "--precision", "bf16", "--no-variants",
                       "--no-cpu-baseline", "--steps", "4", "--warmup", "2"]
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                c5 = json.loads(r.stdout.decode().strip().splitlines()[-1])
                GF5 = 384.4          # algorithmic GFLOP per training pair at T = 80, R = 576
                out.setdefault("variants", {})["cfg5_bf16"] = {
                    "value": c5["value"], "unit": "pairs/s", "ms_per_step": c5["ms_per_step"], "workload": c5["config"]["workload"],
                    "pairs_per_gpu": c5["config"]["pairs_per_gpu"], "regions": c5["config"]["regions"], "dtype": c5["dtype"],
                    "model_tflops": round(c5["value"] * GF5 / 1000.0, 1), "model_frac_of_bf16_peak": round(c5["value"] * GF5 / 1000.0 / PEAK_BF16_MFMA_TFLOPS, 4),
                    "roofline": {k: c5["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us") if k in c5.get("roofline", {})},
                    "families": c5.get("roofline", {}).get("families"),
                    "note": "BASELINE configs[4] per-GPU workload in a separate process (python bench.py --workload cfg5_long_traj_bs32 --precision bf16); not the headline"}
            except Exception as e:
                out.setdefault("variants", {})["cfg5_bf16"] = {"error": f"{type(e).__name__}: {e}"}
            # the END-TO-END step includes the H2D of the batch (utils/utils_init.py:199-207).  Three short child runs of the same
            # workload: the batch uploaded under the previous step (overlap), the compact form expanded on the device, and -- SURVEY 8(f) rank 1 --
            # the loss-aware heads (LM / image logits only where the loss reads them: same losses and gradients).  NOT the headline.
            for tag, extra, note in (("h2d_overlap", ["--h2d", "overlap"], "batch re-uploaded from pinned host memory every step on a copy stream under the previous step"),
                                     ("h2d_compact", ["--h2d", "compact"], "compact batch (distinct frames once, un-masked tokens) uploaded every step, options expanded and masked on the device"),
                                     ("loss_aware_heads", ["--loss-aware-heads"], "decoder / image-head logits only at the positions the losses read (identical losses and gradients)"),
                                     ("reference_loop", ["--loop", "reference"], "the reference's train_epoch body as it stands on the drop-in modules (utils/utils_init.py:199-268): eager launches, "
                                      "model.zero_grad(), every logged scalar read back each step -- what the import swap alone gives")):
                try:
                    import subprocess
                    torch.cuda.empty_cache()
                    cmd = [sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--no-variants", "--no-cpu-baseline", "--no-kernel-timing",
                           "--host-probe", "0", "--steps", "8", "--warmup", "3"] + extra
                    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
                    c = json.loads(r.stdout.decode().strip().splitlines()[-1])
                    out.setdefault("variants", {})[tag] = {"value": c["value"], "unit": "pairs/s", "ms_per_step": c["ms_per_step"], "note": note + "; separate process; not the headline"}
                except Exception as e:
                    out.setdefault("variants", {})[tag] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not dp_wrap and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.workload)
    if rank == 0:
        emit(json.dumps(out))
    if dp_wrap:
        runner.close()
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
