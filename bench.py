#!/usr/bin/env python3
"""Benchmark of the Permutect artifact-model hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a process group in the environment: bench.py starts `python -m torch.distributed.run` with N ranks itself
(before anything touches a GPU) and relays rank 0's line; the driver's own torchrun launch works as before.

Metric (BASELINE.json): read-sets/sec, "train fwd+bwd; filter fwd".  The headline `value` is BASELINE.json configs[1]:
train_model on synthetic WGS-shaped ReadSet batches with the production-shaped hyperparameters P0 (59 845 parameters):
one "step" = forward + losses + backward + (DP: bucketed RCCL gradient all-reduce, the big bucket under the rest of the
backward) + global-norm clip + AdamW on one prepared batch already resident in HBM.  The SAME run then times the
filter_variants forward (configs[2]) over the same resident batches and prints it as `"filter": {...}` inside the line.

ONE JSON line on rank 0 carries, besides the contract's keys:
  roofline      dominant kernel (pmt_backward_kernel), HIP-event timed inside the timed region; `traffic` only from a
                profiles/*.json whose recorded hash of the kernel sources equals this tree's
  filter        the filter-forward half of the metric: value, ms_per_step, its own roofline (pmt_forward_kernel)
  small_batch   the reference's default batch sizes (training 64, inference 8192: parameters.py:214,236-242) and twice the
                headline's batch (131072), N = 1 only
  stress        BASELINE configs[4] on one GPU: 1 420 read sets of ~600 reads per step (the same reads per step as the headline),
                train and filter, each with its own roofline and its per-read rate relative to the WGS path; N = 1 only
  loader        BASELINE configs[1] / [2] as DATASETS: 2^20-variant training epochs and a 5 x 2^20-candidate filter pass (with and
                without the posterior hand-off) streamed through the device chunk loader, H2D inclusive, and each rate's ratio
                to the resident rate; N = 1 only
  parity_check  EVERY variant of resident batch 0 (one whole 65 536-set launch) against the CPU oracle, after the timed
                regions: max / 99.99th percentile / count above the 1e-4 contract; the run FAILS above 1e-4
  cpu_baseline  the CPU oracle (PyTorch-CPU restatement of the reference path) on this box's host cores, thread count
                swept and the best reported, B = 8192 and B = 64, train and filter; N = 1 only
"""
import argparse
import glob
import hashlib
import json
import os
import warnings
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


def note(msg):
    """Progress on stderr (rank 0): the JSON line stays alone on stdout, and a long CPU leg never looks like a hang."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # the same guide: bf16 / f16 MFMA, dense (the 5 PF figure includes 2:1 sparsity): the pipe the kernels issue on
MFMAS_PER_PRODUCT = 3  # an fp32-grade product = three 16-bit MFMAs on two-piece splits of both operands (forward f16, backward bf16)
MFMA_16x16x32_FLOPS = 2 * 16 * 16 * 32
SIMDS, SHADER_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; the nominal shader clock SQ_VALU_MFMA_BUSY_CYCLES is set against
PEAK_HBM_GBS = 8000.0


def synth_arrays(rng, num_variants, depth):
    """SURVEY.md 8d synthetic inputs: packed 12-byte read rows, N(0,1) info, U{0..4} haplotypes, cycling labels."""
    if depth == "wgs":
        nref, nalt = rng.integers(0, 11, num_variants), rng.integers(1, 16, num_variants)
    elif depth == "stress":  # BASELINE configs[4]: mean 600 reads / variant (300x tumor + normal); read sets span several
        nref, nalt = rng.poisson(300, num_variants), np.maximum(rng.poisson(300, num_variants), 1)  # workgroups (layered launches)
    else:  # high depth inside one workgroup per read set: mean 200 reads / variant
        nref, nalt = np.minimum(rng.poisson(100, num_variants), 120), np.clip(rng.poisson(100, num_variants), 1, 120)
    ints = np.zeros((num_variants, 16 + 42), dtype=np.int16)
    ints[:, 0], ints[:, 1] = nref, nalt
    ints[:, 2] = np.arange(num_variants) % 3
    ints[:, 16:] = rng.integers(0, 5, (num_variants, 42))
    floats = np.zeros((num_variants, 6 + 71), dtype=np.float16)
    floats[:, 6:] = rng.standard_normal((num_variants, 71)).astype(np.float16)
    packed = rng.integers(0, 256, (int(nref.sum() + nalt.sum()), 12), dtype=np.uint8)
    return ints, floats, packed


def kernels_sha():
    """Hash of everything the device code is built from: a PMC measurement describes ONE build of the kernels."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "permutect_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "permutect_amd", "csrc", "*.hpp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "permutect_amd", "csrc", "Makefile")])  # (+ the build flags)
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_parity_sweep():
    """The 2^20-variant filter sweep of tests/test_scale_gpu.py as last committed under profiles/ (NOT measured by this run): how many
    variants per million lie beyond 1e-4 of the fp32 oracle, and the fp64 yardstick for them."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_errors.jsonl"))):
        with open(path) as f:
            for ln in f:
                try:
                    rec = json.loads(ln)
                except ValueError:
                    continue
                if rec.get("test") == "filter_sweep_1M":
                    best = {"source": os.path.relpath(path, ROOT) + " (tests/test_scale_gpu.py; a committed measurement, not this run)",
                            "read_sets": rec.get("read_sets"), "logit_errs_over_1e4": rec.get("logit_errs_over_1e4"),
                            "logit_errs_over_1e4_per_million": 1e6 * rec.get("logit_errs_over_1e4", 0) / max(1, rec.get("read_sets", 1)),
                            "max_logit_err": rec.get("max_logit_err"), "max_hip_vs_fp64": rec.get("max_hip_vs_fp64"),
                            "max_fp32_oracle_vs_fp64": rec.get("max_fp32_oracle_vs_fp64")}
    return best


def pmc_record(kernel, batch, depth):
    """The committed rocprofv3 --pmc record of `kernel` (profiles/*pmc_traffic*.json, written by scripts/pmc_traffic.py): HBM bytes per
    launch and, when the SQ pass was given to the script, what the matrix pipe did (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, ...).  The
    same staleness rule as pmc_traffic: this build of the kernels, this workload, or None."""
    sha = kernels_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("kernels_sha") != sha or rec.get("batch_read_sets") != batch or rec.get("depth") != depth:
            continue
        k = rec.get("kernels", {}).get(kernel)
        if k is not None:
            return dict(k, source=os.path.relpath(path, ROOT))
    return None


def pmc_traffic(kernel, batch, depth):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 --pmc measurement (profiles/*pmc_traffic*.json, written by
    scripts/pmc_traffic.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on
    gfx950, + WRITE_SIZE).  Only a record taken on THIS build of the kernels (same hash of csrc/ + include/) and this
    workload counts; anything else is stale and yields None."""
    sha = kernels_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("kernels_sha") != sha or rec.get("batch_read_sets") != batch or rec.get("depth") != depth:
            continue
        k = rec.get("kernels", {}).get(kernel)
        if k is not None:
            return k["hbm_bytes_per_launch"]
    return None


def algorithmic_macs_per_read(model, padded=False):
    """Forward MACs per read of the read-set path: read MLP + L gated blocks + reducer + rotation.
    `padded`: with every dimension rounded up to the 16-wide MFMA tile (what the matrix core actually executes)."""
    d = model.engine().plan.desc
    up = (lambda n: (n + 15) // 16 * 16) if padded else (lambda n: n)
    used = set()

    def mlp(m):
        for i in range(m.n_ops):
            o = m.ops[i]
            for k in range(o.n_layers):
                used.add(o.lin[k])
    mlp(d.read_mlp)
    mlp(d.reducer)
    used.add(d.rotation_lin)
    macs = sum(up(d.lin[i].in_dim) * up(d.lin[i].out_dim) for i in used)
    h = d.d_ffn // 2
    macs += d.num_blocks * (up(d.d_model) * (up(h) + up(h)) + up(h) * up(d.d_model))  # one side's proj1 + proj2 per read
    return macs


# ---- the CPU baseline ---------------------------------------------------------------------------------------------------
def _oracle_problem(b):
    from oracle import artifact_oracle as O  # checker / baseline only -- never the product path
    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.parameters import P0_DIMS, p0_params
    from tests.helpers import config_for
    torch.manual_seed(0)
    ref_model = ArtifactModel(p0_params(), device=torch.device("cpu"), **P0_DIMS)
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    ints, floats, packed = synth_arrays(np.random.default_rng(1), b, "wgs")
    batch = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                 nref=torch.from_numpy(ints[:, 0].astype(np.int64)), nalt=torch.from_numpy(ints[:, 1].astype(np.int64)),
                 labels=torch.from_numpy(ints[:, 2].astype(np.int64)), sources=torch.zeros(b, dtype=torch.int64),
                 info_be=torch.from_numpy(floats[:, 6:].astype(np.float32)),
                 haplotypes_bh=torch.from_numpy(ints[:, 16:].astype(np.int64)))
    return O, config_for("p0"), sd, batch, packed.shape[0]


def _oracle_rate(mode, b, threads, seconds, min_steps=2):
    """read-sets/s of the oracle's train step / filter forward at batch b on `threads` host threads, timed for ~`seconds`."""
    O, cfg, sd, batch, _ = _oracle_problem(b)
    names = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(".base")]
    m = [torch.zeros_like(sd[n]) for n in names]
    v = [torch.zeros_like(sd[n]) for n in names]
    torch.set_num_threads(threads)

    def step(i):
        if mode == "train":
            _, _, grads = O.train_step_grads(sd, cfg, batch)
            O.clip_and_adamw([sd[n] for n in names], [grads[n] for n in names], m, v, step=i + 1, lr=1e-3, weight_decay=0.01)
        else:
            with torch.inference_mode():
                O.compute_batch_output(sd, cfg, batch["reads_re"], batch["nref"], batch["nalt"], batch["info_be"], batch["haplotypes_bh"])

    step(0)
    t0, n = time.perf_counter(), 0
    while n < min_steps or time.perf_counter() - t0 < seconds:
        step(n + 1)
        n += 1
    return b * n / (time.perf_counter() - t0), n


def usable_cpus():
    """Host threads this process can really run at once: the smallest of the machine's cores, its affinity mask and its
    cgroup CPU quota (a GPU box hands a job a share of a 128-core host; OpenMP teams wider than that share spin on each
    other and a step takes minutes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return min(n, 64)


def cpu_baseline():
    """The reference path as restated by oracle/artifact_oracle.py on the host cores.  ~18 000 small ATen ops per step do not
    scale with threads (128 threads on them is oversubscription), so the thread count is swept on the B = 8192 train step and
    the best one is used for the other three points.  Bounded: about 25 s of CPU work in all."""
    ncores = usable_cpus()
    before = torch.get_num_threads()
    candidates = sorted({t for t in (4, 8, 16, 32, 64, ncores) if t <= ncores})
    sweep = {}
    for t in candidates:
        sweep[t], _ = _oracle_rate("train", 8192, t, seconds=1.0)
        note(f"cpu baseline: train B=8192 on {t} threads: {sweep[t]:.0f} read-sets/s")
        if sweep[t] < 0.6 * max(sweep.values()):  # past the knee more threads only add hand-off cost (and, beyond the
            break                                  # process's CPU share, spinning OpenMP workers that crawl)
    best = max(sweep, key=sweep.get)
    train8k, n1 = _oracle_rate("train", 8192, best, seconds=5.0)
    filt8k, n2 = _oracle_rate("filter", 8192, best, seconds=3.0)
    note(f"cpu baseline: B=8192 on {best} threads: train {train8k:.0f}, filter {filt8k:.0f} read-sets/s")
    t64 = min(best, 8)  # 812 reads per step: more threads only add hand-off cost
    train64, n3 = _oracle_rate("train", 64, t64, seconds=3.0, min_steps=10)
    filt64, n4 = _oracle_rate("filter", 64, t64, seconds=2.0, min_steps=10)
    note(f"cpu baseline: B=64 on {t64} threads: train {train64:.0f}, filter {filt64:.0f} read-sets/s")
    torch.set_num_threads(before)
    reads = _oracle_problem(8192)[4]
    return {"value": train8k, "unit": "read-sets/s", "cores": best, "kind": "port",
            "sample": f"{n1} train steps of B=8192 WGS-shaped read sets ({reads} reads), P0, fp32, oracle/artifact_oracle.py "
                      f"(PyTorch-CPU restatement of the reference path, fwd + losses + autograd bwd + clip + AdamW); thread "
                      f"count swept over {sorted(sweep)} (usable host threads: {ncores} of {os.cpu_count()} cores), best = {best}",
            "host_cores": ncores, "thread_sweep_train_b8192": {str(k): v for k, v in sweep.items()},
            "filter": {"value": filt8k, "unit": "read-sets/s", "cores": best, "sample": f"{n2} forwards of B=8192 under inference_mode"},
            "b64": {"train": train64, "filter": filt64, "unit": "read-sets/s", "cores": t64,
                    "sample": f"{n3} train steps / {n4} forwards of B=64 (the reference's default training batch)"}}


# ---- launching ----------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """`python bench.py --gpus N` without a process group in the environment: run N ranks under torch.distributed.run as a
    child process (this process has not touched a GPU and never will) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        if line.startswith("{"):
            print(line, flush=True)
    raise SystemExit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~1 s of train steps, ~0.2 s of filter forwards
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["both", "train", "filter"], default="both",
                    help="both: the headline is the train step and the filter forward is timed in the same run (\"filter\" key); "
                         "train / filter: only that half (filter: the headline keys describe the filter forward)")
    ap.add_argument("--batch", type=int, default=65536, help="read sets per step per GPU")
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct synthetic batches kept in HBM per GPU")
    ap.add_argument("--depth", choices=["wgs", "high", "stress"], default="wgs")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the reference's arithmetic and the parity headline): fp32-equivalent products (six bf16 MFMAs on "
                         "three-piece splits / exact fp32 MFMAs).  bf16: ONE bf16 MFMA per product on single roundings of both operands -- "
                         "BASELINE.json's 'bf16' training configuration, which the reference itself never computes in; a separate, labelled "
                         "line with its measured logit error, no parity claim")
    ap.add_argument("--preroll", type=float, default=0.25, help="seconds of untimed steps before the counted warm-up of the headline loops (clock ramp)")
    ap.add_argument("--strict-parity", action="store_true",
                    help="fail on ANY variant beyond 1e-4 of the fp32 oracle; default: such a variant is recomputed in fp64 and the run fails "
                         "unless the excess is the fp32 reference's own rounding (counted and reported in parity_check either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the small-batch points and the oracle parity check")
    ap.add_argument("--rehearse", action="store_true",
                    help="development only: run the N > 1 code path on a box with fewer GPUs than ranks (ranks share the "
                         "cards, gloo instead of RCCL); the numbers mean nothing")
    ap.add_argument("--data", choices=["resident", "loader"], default="resident",
                    help="resident: batches already in HBM (the metric's definition); loader: batches drawn from a "
                         "synthetic dataset in host memory through the device chunk loader, chunk uploads inside the timed "
                         "region (the H2D-inclusive rate of SURVEY 8d; reported in DESIGN.md, never the headline value)")
    ap.add_argument("--dataset-variants", type=int, default=1 << 20, help="--data loader: variants in the synthetic dataset")
    ap.add_argument("--chunk-variants", type=int, default=1 << 18, help="--data loader: variants per HBM-resident chunk")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)  # does not return
    if args.dtype == "bf16":
        os.environ["PMT_SHAPE"] = "bf16"  # read once, when the model is lowered (engine/plan.py): PmtModel.force_shape = 3
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from permutect_amd.architecture.artifact_model import ArtifactModel
    from permutect_amd.data.batch import Batch
    from permutect_amd.parameters import P0_DIMS, p0_params
    from permutect_amd.training.distributed import BucketedGradAllReduce
    from permutect_amd.training.optimizer import FusedClipAdamW

    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(0)  # identical initial weights on every rank
    model = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    eng = model.engine()
    reduce_grads = None
    if dist is not None:
        dist.broadcast(eng.space.theta, src=0)
        eng.params_changed()
        reduce_grads = BucketedGradAllReduce()  # loss is a batch SUM (reference artifact_model.py:90) -> SUM all-reduce
        eng.grad_hook = reduce_grads            # the early bucket goes out under the haplotype-CNN / info-MLP backward

    rng = np.random.default_rng(1000 + rank)  # each rank owns a different shard of the synthetic dataset
    batches, reads_total = [], 0
    stream_batches = None
    first_host = None
    if args.data == "resident":
        for i in range(args.resident_batches):
            ints, floats, packed = synth_arrays(rng, args.batch, args.depth)
            if i == 0:
                first_host = (ints, floats, packed)
            b = Batch.from_arrays(ints, floats, packed, pack=True)  # variants in the order that fills the workgroups best
            b.plan(allow_split=True)
            batches.append(b.copy_to(dev))
            reads_total += packed.shape[0]
        reads_per_batch = reads_total / len(batches)
        groups_per_batch = float(np.mean([b.plan(allow_split=True).num_groups for b in batches]))
    else:
        from permutect_amd.data.memory_mapped_data import MemoryMappedData
        from permutect_amd.data.reads_dataset import ReadsDataset
        ints, floats, packed = synth_arrays(rng, args.dataset_variants, args.depth)
        # on-disk order: per datum its ref rows then its alt rows (synth rows are i.i.d., so the order is immaterial)
        dataset = ReadsDataset(MemoryMappedData.from_arrays(ints, floats, packed))
        reads_per_batch = packed.shape[0] / args.dataset_variants * args.batch
        groups_per_batch = None
        loader_shuffle = [True]

        def endless():
            while True:
                # (training draws shuffled batches; filter_variants walks the candidates in order)
                for cb in dataset.device_loader(args.batch, dev, chunk_variants=args.chunk_variants, rng=rng, shuffle=loader_shuffle[0]):
                    if cb.size() == args.batch:
                        yield cb
        stream_batches = endless()
    torch.cuda.synchronize()
    note(f"{len(batches)} resident batches of {args.batch} read sets built")

    def train_step(batch):
        opt.zero_grad()
        out = model.compute_batch_output(batch)
        losses = model.compute_batch_losses(out, batch)
        losses.total_loss.backward()
        opt.step(pre_reduce=reduce_grads)
        return out, losses

    def filter_step(batch):
        with torch.inference_mode():
            return model.compute_batch_output(batch)

    prerolled = {}
    preroll_s = args.preroll

    def timed(mode, pool, steps, warmup):
        """W untimed + exactly K timed steps between barrier + synchronize brackets; max over ranks.  Returns (seconds,
        kernel milliseconds, per-step milliseconds): the last from one HIP event behind every step -- no synchronisation
        inside the loop -- so that the spread of the steps is visible next to their mean."""
        model.train(mode == "train")
        fn = train_step if mode == "train" else filter_step
        nxt = (lambda i: pool[i % len(pool)]) if stream_batches is None or pool is not batches else (lambda i: next(stream_batches))
        # untimed pre-roll on top of the counted warm-up: the card's clocks need ~15 ms of work to come back after an idle stretch and a
        # `--steps 20` run would sit inside that ramp; resident batches only (a streamed pass has its own fill)
        t_pre = time.perf_counter()
        did_preroll = preroll_s > 0 and pool is batches and stream_batches is None
        if did_preroll:
            go = True
            while go:
                for i in range(4):
                    fn(nxt(i))
                torch.cuda.synchronize()
                go = time.perf_counter() - t_pre < preroll_s
                if dist is not None:  # every rank takes the SAME number of steps (each one is a collective): rank 0's clock decides
                    flag = torch.tensor([1 if go else 0], device=dev, dtype=torch.int32)
                    dist.broadcast(flag, src=0)
                    go = bool(flag.item())
        if did_preroll:
            prerolled[mode] = time.perf_counter() - t_pre
        for i in range(warmup):
            fn(nxt(i))
        eng.timers = {"pmt_forward": [], "pmt_backward": []}
        eng.timer_stride, eng._timer_calls = (4 if steps >= 16 else 1), {}  # HIP events around every 4th launch of the timed region
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            fn(nxt(warmup + i))
            marks[i + 1].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kernel_ms = {k: (sum(s.elapsed_time(e) for s, e in v) / len(v) if v else None) for k, v in eng.timers.items()}
        kernel_ms["_samples"] = {k: len(v) for k, v in eng.timers.items()}
        eng.timers, eng.timer_stride = None, 1
        per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)])
        return elapsed, kernel_ms, per_step

    def step_stats(elapsed, per_step):
        return {"timed_s": elapsed, "step_ms_min": float(per_step.min()), "step_ms_median": float(np.median(per_step)),
                "step_ms_max": float(per_step.max())}

    macs = algorithmic_macs_per_read(model)
    pad_ratio = algorithmic_macs_per_read(model, padded=True) / macs
    fwd_flops = 2.0 * macs * reads_per_batch

    def roofline(kernel, flops, ms):
        """`peak` / `frac`: the dense fp32 MFMA rate -- the roof of the ARITHMETIC the contract asks for (fp32-grade products), and the
        scale earlier rounds are quoted on.  It is not the roof of the pipe the kernels issue on: they run three 16-bit MFMAs per
        product, so the scheme's own ceiling is pipe_peak / 3 fp32-equivalent TFLOP/s (`scheme_ceiling`, `frac_of_scheme_ceiling`),
        and what the matrix pipe really did comes from the committed SQ counters of this build (`executed_mfma_flops`: every MFMA
        issued, pieces, tile padding and in-kernel recomputation included; `mfma_busy_frac`: the pipe's busy cycles over SIMDs x launch
        time x the nominal clock).  The counters are a profiled run's (scripts/collect_profiles.sh); None when no record matches this
        tree's kernels."""
        achieved = flops / (ms * 1e-3) / 1e12
        pmc = pmc_record(kernel, args.batch, args.depth) or {}
        traffic = pmc.get("hbm_bytes_per_launch")
        ceiling = PEAK_16BIT_MFMA_TFLOPS / MFMAS_PER_PRODUCT
        n_mfma, busy = pmc.get("SQ_INSTS_MFMA"), pmc.get("SQ_VALU_MFMA_BUSY_CYCLES")
        wait, wave = pmc.get("SQ_WAIT_ANY"), pmc.get("SQ_WAVE_CYCLES")
        return {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "kernel": kernel, "kernel_ms": ms,
                "algorithmic_flops_per_launch": flops, "padded_flops_per_launch": flops * pad_ratio,
                "frac_padded": achieved * pad_ratio / PEAK_FP32_MFMA_TFLOPS,
                "hbm_frac": None if traffic is None else traffic / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9),
                "pipe_peak": PEAK_16BIT_MFMA_TFLOPS, "mfmas_per_product": MFMAS_PER_PRODUCT, "scheme_ceiling": ceiling,
                "frac_of_scheme_ceiling": achieved / ceiling,
                "executed_mfma_flops": None if n_mfma is None else n_mfma * MFMA_16x16x32_FLOPS,
                "executed_over_algorithmic": None if n_mfma is None else n_mfma * MFMA_16x16x32_FLOPS / flops,
                "mfma_busy_frac": None if busy is None else busy / (SIMDS * ms * 1e-3 * SHADER_GHZ * 1e9),
                "wait_any_frac": None if not (wait and wave) else wait / wave,
                "pmc_source": pmc.get("source")}

    results = {}
    if args.mode in ("both", "train"):
        elapsed, kms, per_step = timed("train", batches, args.steps, args.warmup)
        r = roofline("pmt_backward_kernel", 2.0 * fwd_flops, kms["pmt_backward"])  # dgrad + wgrad; in-kernel recompute not counted
        r["matrix_pipe"] = ("fp32 accumulation on the 16-bit matrix pipes.  Forward (both kernels): fp32-equivalent, three f16 MFMAs on two-piece "
                            "splits of both operands with the low piece scaled by 2^12 (23 significant bits per operand; six bf16 MFMAs on "
                            "three-piece splits until round 3, PMT_SHAPE=bf16x3).  Backward: dgrad, recompute and wgrad as three bf16 MFMAs on "
                            "two-piece splits (16 significant bits per operand, `bwd_operand_bits`): gradients 7 - 8e-6 from an fp64 evaluation, "
                            "the reference's own fp32 arithmetic 7 - 9e-6 (tests/test_scale_gpu.py); over a 100-step trajectory no farther from "
                            "fp64 than the fp32 reference or the six-MFMA build (tests/test_trajectory_gpu.py); `peak` is the dense fp32 MFMA rate"
                            if args.dtype == "f32" else
                            "plain bf16: one bf16 MFMA per product (fp32 accumulation), single roundings of both operands; `peak` stays the "
                            "dense fp32 MFMA rate so that the two modes read on one scale (the bf16 dense peak is ~2.5 PFLOP/s)")
        r["other_kernel_ms"] = kms
        r["kernel_timing"] = f"HIP events on the launch stream around {kms['_samples']['pmt_backward']} of the {args.steps} launches inside the timed region"
        results["train"] = (elapsed, r, step_stats(elapsed, per_step))
        note(f"train: {1e3 * elapsed / args.steps:.3f} ms/step over {elapsed:.3f} s (steps {per_step.min():.3f} .. {per_step.max():.3f} ms), kernels {kms}")
    if args.mode in ("both", "filter"):
        if stream_batches is not None:
            loader_shuffle[0] = False
        elapsed, kms, per_step = timed("filter", batches, args.steps, args.warmup)
        rf_ = roofline("pmt_forward_kernel", fwd_flops, kms["pmt_forward"])
        rf_["kernel_timing"] = f"HIP events on the launch stream around {kms['_samples']['pmt_forward']} of the {args.steps} launches inside the timed region"
        results["filter"] = (elapsed, rf_, step_stats(elapsed, per_step))
        note(f"filter: {1e3 * elapsed / args.steps:.3f} ms/step over {elapsed:.3f} s (steps {per_step.min():.3f} .. {per_step.max():.3f} ms), kernels {kms}")

    # ---- the reference's default batch sizes beside the build's best: N = 1, resident batches only ------------
    small = None
    if world == 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs":
        small = {}
        for bsz, k in ((64, 200), (8192, 40), (131072, 10)):  # (the last: twice the headline's batch)
            pool = []
            srng = np.random.default_rng(77 + bsz)
            for _ in range(4 if bsz <= 8192 else 2):
                b = Batch.from_arrays(*synth_arrays(srng, bsz, "wgs"), pack=True)
                b.plan(allow_split=True)
                pool.append(b.copy_to(dev))
            et, _, _ = timed("train", pool, k, 10)
            ef, _, _ = timed("filter", pool, k, 10)
            if bsz > 8192:
                note(f"B={bsz}: train {1e3 * et / k:.3f} ms/step, filter {1e3 * ef / k:.3f} ms/step")
                small[f"b{bsz}"] = {"train_read_sets_per_s": bsz * k / et, "train_ms_per_step": 1e3 * et / k,
                                    "filter_read_sets_per_s": bsz * k / ef, "filter_ms_per_step": 1e3 * ef / k}
                continue
            hosts = [Batch.from_arrays(*synth_arrays(srng, bsz, "wgs"), pack=True).pin_memory() for _ in range(4)]
            # like for like: the EAGER step with the same per-step upload of a pinned host batch
            model.train(True)
            for i in range(10):
                train_step(hosts[i % 4].copy_to(dev))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                train_step(hosts[i % 4].copy_to(dev))
            torch.cuda.synchronize()
            eu = time.perf_counter() - t0
            note(f"B={bsz}: train {1e3 * et / k:.3f} ms/step resident, {1e3 * eu / k:.3f} with a batch upload per step; filter {1e3 * ef / k:.3f} ms/step")
            small[f"b{bsz}"] = {"train_read_sets_per_s": bsz * k / et, "train_ms_per_step": 1e3 * et / k,
                                "train_with_upload_ms_per_step": 1e3 * eu / k,
                                "filter_read_sets_per_s": bsz * k / ef, "filter_ms_per_step": 1e3 * ef / k}

    # ---- BASELINE configs[4]: high-depth stress, mean 600 reads per variant (read sets split over workgroups: layered launches) --
    stress = None
    if world == 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both":
        sb = 1420  # the same ~852 K reads per step as the WGS headline batch
        pool, sreads = [], 0
        srng = np.random.default_rng(4040)
        for _ in range(2):
            si, sf, sp = synth_arrays(srng, sb, "stress")
            b = Batch.from_arrays(si, sf, sp, pack=True)
            b.plan(allow_split=True)
            pool.append(b.copy_to(dev))
            sreads += sp.shape[0]
        sreads /= len(pool)
        k = 30
        et, kt, pt = timed("train", pool, k, 6)
        ef, kf, pf = timed("filter", pool, k, 6)
        sflops = 2.0 * macs * sreads
        wgs_train_reads_per_s = reads_per_batch * args.steps / results["train"][0]
        wgs_filter_reads_per_s = reads_per_batch * args.steps / results["filter"][0]

        def sroof(kernel, flops, ms, launches):
            ach = flops / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                    "traffic": None, "kernel": kernel, "kernel_ms": ms, "launches_per_step": launches, "algorithmic_flops_per_step": flops}
        nl = 1 if eng.join_layered else eng.plan.desc.num_blocks + 1
        how = ("read sets split over workgroups and JOINED inside one launch each way (per-set sums through HBM + arrival counters, activations in registers)"
               if eng.join_layered else f"layered execution ({nl} launches each way, activations parked in HBM in between)")
        torch.cuda.synchronize()
        eng.check_join_fault()  # (a joined launch that gave up waiting would have produced wrong numbers: fail loudly)
        stress = {"workload": f"BASELINE configs[4]: {sb} read sets per step, mean {sreads / sb:.0f} reads per set ({sreads:.0f} reads per step); {how}",
                  "train": {"value": sb * k / et, "unit": "read-sets/s", "ms_per_step": 1e3 * et / k, **step_stats(et, pt), "reads_per_s": sreads * k / et,
                            "per_read_rate_vs_wgs": (sreads * k / et) / wgs_train_reads_per_s,
                            "roofline": sroof("pmt_backward_kernel<ShapeP0X, split read sets>", 2.0 * sflops, kt["pmt_backward"], nl)},
                  "filter": {"value": sb * k / ef, "unit": "read-sets/s", "ms_per_step": 1e3 * ef / k, **step_stats(ef, pf), "reads_per_s": sreads * k / ef,
                             "per_read_rate_vs_wgs": (sreads * k / ef) / wgs_filter_reads_per_s,
                             "roofline": sroof("pmt_forward_kernel<false, ShapeP0X, split read sets>", sflops, kf["pmt_forward"], nl)}}
        note(f"stress: train {1e3 * et / k:.3f} ms/step ({stress['train']['per_read_rate_vs_wgs']:.2f} x the WGS per-read rate), "
             f"filter {1e3 * ef / k:.3f} ms/step ({stress['filter']['per_read_rate_vs_wgs']:.2f} x)")
        del pool

    # ---- the same training step with the reference's --dropout_p (mlp.py:57-58): the masks are generated in the kernels, such a
    #      step runs the generic instances ("Dropout") ---------------------------------------------------------
    dropout = None
    if world == 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both":
        dparams = p0_params()
        dparams.dropout_p = 0.25
        dmodel = ArtifactModel(dparams, device=dev, **P0_DIMS)
        dopt = FusedClipAdamW(dmodel, lr=1e-3, weight_decay=0.01)
        dmodel.train(True)

        def dstep(batch):
            dopt.zero_grad()
            out = dmodel.compute_batch_output(batch)
            dmodel.compute_batch_losses(out, batch).total_loss.backward()
            dopt.step()
        for i in range(4):
            dstep(batches[i % len(batches)])
        k = 20
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        marks[0].record()
        for i in range(k):
            dstep(batches[i % len(batches)])
        marks[1].record()
        torch.cuda.synchronize()
        dms = marks[0].elapsed_time(marks[1]) / k
        dropout = {"workload": "the headline training step with dropout_p = 0.25 (a new mask seed per step; generic kernel instances)",
                   "ms_per_step": dms, "value": args.batch / (dms * 1e-3), "unit": "read-sets/s", "steps": k,
                   "vs_no_dropout": (results["train"][0] / args.steps * 1e3) / dms}
        note(f"dropout_p = 0.25: train {dms:.3f} ms/step ({dropout['vs_no_dropout']:.2f} x the rate without dropout)")
        del dmodel, dopt

    # ---- BASELINE configs[1] / [2] as DATASETS: batches streamed from host memory through the device chunk loader -------------
    loader = None
    if world == 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both":
        from permutect_amd.data.memory_mapped_data import MemoryMappedData
        from permutect_amd.data.reads_dataset import ReadsDataset
        from permutect_amd.tools.posterior_data import make_posterior_mmap
        t0 = time.perf_counter()
        lrng = np.random.default_rng(5050)
        li, lf, lp = synth_arrays(lrng, 1 << 20, "wgs")
        # the tools' own path: train_artifact_model / make_posterior_mmap page-lock a dataset that fits host memory (ReadsDataset.
        # pin_memory_if_it_fits; PMT_PIN_DATASET=0 or a dataset beyond memory: staged copies), and so does this section
        ds1 = ReadsDataset(MemoryMappedData.from_arrays(li, lf, lp))
        pinned = ds1.pin_memory_if_it_fits()
        note(f"loader: 2^20-variant dataset, page-locked: {pinned} ({time.perf_counter() - t0:.1f} s)")
        bsz, chunk = args.batch, 1 << 18

        def epochs(shuffle):
            while True:
                for cb in ds1.device_loader(bsz, dev, chunk_variants=chunk, rng=lrng, shuffle=shuffle):
                    if cb.size() == bsz:
                        yield cb

        def timed_stream(mode, gen, steps, warmup):
            model.train(mode == "train")
            fn = train_step if mode == "train" else filter_step
            for _ in range(warmup):
                fn(next(gen))
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                fn(next(gen))
            torch.cuda.synchronize()
            return time.perf_counter() - t
        per_epoch = (1 << 20) // bsz
        k_train = 2 * per_epoch
        et = timed_stream("train", epochs(True), k_train, per_epoch)          # one epoch of warm-up, two timed: 2^21 variants
        train_rate = bsz * k_train / et
        # filter_variants over 5 x 2^20 candidates, INCLUDING the posterior hand-off (tools/posterior_data.make_posterior_mmap:
        # rows back on the host in dataset order): the five-fold dataset repeats the 2^20 synthetic variants
        ds5 = ReadsDataset(MemoryMappedData.from_arrays(np.concatenate([li] * 5), np.concatenate([lf] * 5), np.concatenate([lp] * 5)))
        pinned = ds5.pin_memory_if_it_fits() and pinned
        n5 = len(ds5)
        make_posterior_mmap(ds1, model, bsz, chunk_variants=chunk)               # warm-up pass over 2^20
        k_filter = n5 // bsz
        model.train(False)
        gen5 = (cb for cb in ds5.device_loader(bsz, dev, chunk_variants=chunk, shuffle=False))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for cb in gen5:
            filter_step(cb)
        torch.cuda.synchronize()
        ef = time.perf_counter() - t
        t = time.perf_counter()
        post = make_posterior_mmap(ds5, model, bsz, chunk_variants=chunk)
        ep = time.perf_counter() - t
        resident_train = args.batch * args.steps / results["train"][0]
        resident_filter = args.batch * args.steps / results["filter"][0]
        # ---- the loop the CLI runs: training/model_training.train_artifact_model itself on the 2^20 dataset --
        # every parent batch downsampled twice on the device, balancer weights, loss recorder, ONE host sync per epoch.  Three training
        # epochs (the first one warms up: allocator, clocks, loader threads), timed by the epoch lines the loop logs right after its
        # per-epoch sync; no validation set, no evaluation pass; the downsampler's balance fit (a one-off ~15 s of CPU work in front of
        # the loop) is skipped: uniform mixture weights.
        from permutect_amd.parameters import TrainingParameters
        from permutect_amd.training.model_training import train_artifact_model
        emodel = ArtifactModel(p0_params(), device=dev, **P0_DIMS)
        stamps = []
        train_artifact_model(emodel, ds1, None, TrainingParameters(batch_size=bsz, num_epochs=3, learning_rate=1e-3, weight_decay=0.01, fit_downsampler=False),
                             chunk_variants=chunk, seed=9, log=lambda msg: stamps.append((time.perf_counter(), msg)), evaluate_every_epoch=False)
        epoch_s = [b[0] - a[0] for a, b in zip(stamps, stamps[1:])]  # epochs 2 and 3
        steps_per_epoch = 2 * (-(-(1 << 20) // bsz))
        epoch_rate = steps_per_epoch * bsz / min(epoch_s)
        epoch = {"what": "training.model_training.train_artifact_model (the loop tools/train_artifact_model.py runs) on the 2^20-variant dataset: two device "
                         "downsamplings of every parent batch, balancer, loss recorder, one host sync per epoch; H2D inclusive; read sets counted as "
                         "optimizer-step batches (2 per parent batch, like the reference's loop: model_training.py:153); best of epochs 2 and 3",
                 "value": epoch_rate, "unit": "read-sets/s", "epoch_s": epoch_s, "steps_per_epoch": steps_per_epoch,
                 "ms_per_step": 1e3 * min(epoch_s) / steps_per_epoch, "vs_resident": epoch_rate / resident_train,
                 "note": "a downsampled batch keeps a Beta-distributed fraction of its parent's reads, so a loop step holds fewer reads than a resident "
                         "bench step of the same read-set count; vs_resident compares read sets per second"}
        note(f"epoch loop: {epoch['ms_per_step']:.3f} ms per optimizer step = {epoch['vs_resident']:.2f} x the resident rate (epochs {epoch_s})")
        # ... and the evaluation pass the loop runs after every validation epoch (reference model_training.py:204-228): three downsamplings of
        # every parent batch, forward with the balancer's weights, the tallies of EvaluationMetrics -- here over the training data only
        from permutect_amd.training.balancer import Balancer
        from permutect_amd.training.downsampler import Downsampler
        from permutect_amd.training.loss_recorder import collect_evaluation_data
        ebal, edown = Balancer(1, dev), Downsampler(1).to(dev)
        for timed_pass in (False, True):
            torch.cuda.synchronize()
            t = time.perf_counter()
            ev = collect_evaluation_data(emodel, ebal, edown, ds1.device_loader(bsz, dev, chunk_variants=chunk, shuffle=False), None, seed=3)
            acc = ev.accuracy(0)  # (the pass's one host sync)
            eval_s = time.perf_counter() - t
        eval_steps = 3 * (-(-(1 << 20) // bsz))
        epoch["evaluation"] = {"what": "collect_evaluation_data over the same dataset: 3 downsamplings per parent batch, filter forward, balancer, tallies; H2D inclusive",
                               "value": eval_steps * bsz / eval_s, "unit": "read-sets/s", "ms_per_step": 1e3 * eval_s / eval_steps, "steps": eval_steps,
                               "accuracy_of_the_untrained_model": acc}
        note(f"evaluation pass: {epoch['evaluation']['ms_per_step']:.3f} ms per forward step ({eval_steps} steps)")
        del emodel
        loader = {"workload": "batches composed on the device from 2^18-variant chunks that the device chunk loader streams out of a synthetic dataset "
                              "in host memory (H2D inside the timed region); page-locked by ReadsDataset.pin_memory_if_it_fits exactly as "
                              "train_artifact_model / make_posterior_mmap do it", "dataset_page_locked": bool(pinned),
                  "train": {"dataset_variants": 1 << 20, "value": train_rate, "unit": "read-sets/s", "ms_per_step": 1e3 * et / k_train, "steps": k_train,
                            "timed_s": et, "vs_resident": train_rate / resident_train},
                  "filter": {"dataset_variants": n5, "value": n5 / ef, "unit": "read-sets/s", "ms_per_step": 1e3 * ef / k_filter, "timed_s": ef,
                             "vs_resident": (n5 / ef) / resident_filter},
                  "filter_with_posterior_handoff": {"dataset_variants": n5, "value": n5 / ep, "unit": "read-sets/s", "timed_s": ep,
                                                    "vs_resident": (n5 / ep) / resident_filter, "rows_out": int(post.num_data),
                                                    "what": "make_posterior_mmap: forward + rows (logit as float16, embedding; integer rows with the counts "
                                                            "zeroed) back in host memory in dataset order"}}
        note(f"loader: train {loader['train']['ms_per_step']:.3f} ms/step = {loader['train']['vs_resident']:.2f} x resident; filter "
             f"{loader['filter']['ms_per_step']:.3f} ms/step = {loader['filter']['vs_resident']:.2f} x resident; with the posterior hand-off "
             f"{n5 / ep / 1e6:.1f} M read-sets/s = {loader['filter_with_posterior_handoff']['vs_resident']:.2f} x")
        del ds1, ds5, post, li, lf, lp

    # ---- N > 1: filter_variants as it is split over GPUs -- the candidates of ONE dataset in N contiguous shards, a rank each,
    #      no collective on the data path, the shards' posterior rows concatenated on rank 0 (tools/posterior_data.make_posterior_mmap) ----
    sharded_filter = None
    if world > 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both":
        try:
            from permutect_amd.data.memory_mapped_data import MemoryMappedData
            from permutect_amd.data.reads_dataset import ReadsDataset
            from permutect_amd.tools.posterior_data import make_posterior_mmap
            n_cand = (1 << 18) * world if not args.rehearse else 1 << 15
            fi, ff, fp = synth_arrays(np.random.default_rng(6060), n_cand, "wgs")  # the SAME dataset on every rank: each takes its shard
            dsf = ReadsDataset(MemoryMappedData.from_arrays(fi, ff, fp))
            fb = min(args.batch, max(1, n_cand // world))
            make_posterior_mmap(dsf, model, fb, chunk_variants=1 << 18, rank=rank, world_size=world)  # warm-up pass
            torch.cuda.synchronize()
            dist.barrier()
            t = time.perf_counter()
            post = make_posterior_mmap(dsf, model, fb, chunk_variants=1 << 18, rank=rank, world_size=world)
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if rank == 0 and (post is None or len(post) != n_cand):
                raise RuntimeError(f"the sharded filter pass returned {None if post is None else len(post)} rows for {n_cand} candidates")
            if rank == 0:
                sharded_filter = {"what": f"make_posterior_mmap over ONE dataset of {n_cand} candidates cut into {world} contiguous shards (a rank each, no "
                                          "collective on the data path), disk-order rows home on rank 0: loader + forward + posterior rows + the "
                                          "concatenation, H2D and the host-side gather inclusive",
                                  "value": n_cand / float(tt.item()), "unit": "read-sets/s", "timed_s": float(tt.item()), "candidates": n_cand, "rows_on_rank0": int(len(post))}
                note(f"sharded filter pass: {n_cand} candidates over {world} ranks in {float(tt.item()):.3f} s")
            del dsf, post, fi, ff, fp
        except Exception as exc:  # noqa: BLE001 -- an extra leg must not take the headline line with it (it has never met a real multi-GPU node)
            sharded_filter = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"} if rank == 0 else None

    # ---- a model that is NOT the production shape: the reference's test configuration T0 on its own kernel instances (engine/
    #      instances.py: the library built around its tile counts) against the generic instance every such model used to run ---------
    shapes = None
    if world == 1 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both" and args.dtype == "f32":
        from permutect_amd.parameters import t0_params, wide_params
        shapes = {}
        for label, env in (("instance", None), ("generic", "any"), ("wide_instance", None), ("wide_generic", "any")):
            make_params = wide_params if label.startswith("wide") else t0_params
            if env is None:
                os.environ.pop("PMT_SHAPE", None)
            else:
                os.environ["PMT_SHAPE"] = env  # read once, when the model is lowered
            torch.manual_seed(1)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # (the generic legs are asked for: the library's warning about them is not news)
                tmodel = ArtifactModel(make_params(), device=dev, **P0_DIMS)
                topt = FusedClipAdamW(tmodel, lr=1e-3, weight_decay=0.01)
                tmodel.engine()  # (lowered here, while the variable is set)
            os.environ.pop("PMT_SHAPE", None)

            def tstep(batch, train):
                if not train:
                    with torch.inference_mode():
                        return tmodel.compute_batch_output(batch)
                topt.zero_grad()
                tmodel.compute_batch_losses(tmodel.compute_batch_output(batch), batch).total_loss.backward()
                topt.step()
            rec = {"pmt_shape_id": tmodel.engine().shape_id, "library_shape": list(__import__("permutect_amd.engine.lib", fromlist=["x"]).shape_of(tmodel.engine().lib))}
            for mode in ("train", "filter"):
                tmodel.train(mode == "train")
                for i in range(4):
                    tstep(batches[i % len(batches)], mode == "train")
                k = 8 if label == "wide_generic" else 20
                marks = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                tmodel.engine().timers = {"pmt_forward": [], "pmt_backward": []}
                torch.cuda.synchronize()
                marks[0].record()
                for i in range(k):
                    tstep(batches[i % len(batches)], mode == "train")
                marks[1].record()
                torch.cuda.synchronize()
                rec[f"{mode}_ms_per_step"] = marks[0].elapsed_time(marks[1]) / k
                for kname, evs in tmodel.engine().timers.items():  # the read-set kernels themselves (the step also holds T0's 64-channel
                    if evs:                                        # haplotype CNN, which runs the general CNN kernels either way)
                        rec[f"{mode}_{kname}_kernel_ms"] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
                tmodel.engine().timers = None
            shapes[label] = rec
            del tmodel, topt
        shapes["workload"] = (f"the reference's test configuration T0 (read_layers [10,10,10], reducer [20,20,20], 2 gated blocks) on the headline's "
                              f"{args.batch}-set batches: the library built around its tile counts against the generic instance; wide_*: the same for "
                              "parameters.wide_params (layers beyond 64: read width 48, d_model 98, d_ffn 32, 2 gated blocks, the production CNN): "
                              "its exact instances with 8-tile register arrays against the generic instances of the wide build")
        note(f"wide shape (d_model 98): train {shapes['wide_instance']['train_ms_per_step']:.2f} vs {shapes['wide_generic']['train_ms_per_step']:.2f} ms, "
             f"filter {shapes['wide_instance']['filter_ms_per_step']:.2f} vs {shapes['wide_generic']['filter_ms_per_step']:.2f} ms (exact instances vs generic)")
        note(f"T0 shape, read-set kernels on its instances vs generic: forward {shapes['instance']['filter_pmt_forward_kernel_ms']:.3f} vs "
             f"{shapes['generic']['filter_pmt_forward_kernel_ms']:.3f} ms, training forward {shapes['instance']['train_pmt_forward_kernel_ms']:.3f} vs "
             f"{shapes['generic']['train_pmt_forward_kernel_ms']:.3f}, backward {shapes['instance']['train_pmt_backward_kernel_ms']:.3f} vs "
             f"{shapes['generic']['train_pmt_backward_kernel_ms']:.3f}; steps {shapes['instance']['train_ms_per_step']:.3f} vs {shapes['generic']['train_ms_per_step']:.3f} (train), "
             f"{shapes['instance']['filter_ms_per_step']:.3f} vs {shapes['generic']['filter_ms_per_step']:.3f} (filter)")

    # ---- the same training step on the build whose backward runs six-MFMA (fp32-equivalent) products: what the 16-bit operands buy ----
    six = None
    alt6 = os.path.join(ROOT, "permutect_amd", "libpermutect_amd_alt6.so")
    if (world == 1 and rank == 0 and not args.no_extras and args.data == "resident" and args.depth == "wgs" and args.mode == "both"
            and args.dtype == "f32" and os.path.exists(alt6) and "PMT_LIB" not in os.environ):
        torch.cuda.synchronize()
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", "train", "--steps", "100", "--warmup", "20", "--no-extras",
                              "--no-cpu-baseline", "--batch", str(args.batch)], env=dict(os.environ, PMT_LIB=alt6), stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode == 0 and lines:
            a6 = json.loads(lines[0])
            six = {"build": "libpermutect_amd_alt6.so: -DPMT_DGRAD_PIECES=3 -DPMT_RECOMPUTE_PIECES=3 (backward products fp32-equivalent)",
                   "ms_per_step": a6["ms_per_step"], "backward_kernel_ms": a6["roofline"]["kernel_ms"],
                   "vs_default_step": a6["ms_per_step"] / (1e3 * results["train"][0] / args.steps)}
            note(f"six-MFMA backward build: {six['ms_per_step']:.3f} ms/step (kernel {six['backward_kernel_ms']:.3f} ms) = {six['vs_default_step']:.3f} x the default step")

    # ---- parity of what was just timed: EVERY variant of resident batch 0 (one whole launch) against the CPU oracle ------------
    parity = None
    if rank == 0 and not args.no_extras and first_host is not None and args.depth == "wgs" and batches:
        from oracle import artifact_oracle as O  # the checker, never the product path
        from tests.helpers import config_for
        ints, floats, packed = first_host
        model.train(False)
        with torch.inference_mode():
            out = model.compute_batch_output(batches[0])
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        i64 = torch.from_numpy(ints.astype(np.int64))
        before = torch.get_num_threads()
        torch.set_num_threads(min(usable_cpus(), 16))
        with torch.inference_mode():
            ref = O.compute_batch_output(sd, config_for("p0"), torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                                         i64[:, 0], i64[:, 1], torch.from_numpy(floats[:, 6:].astype(np.float32)), i64[:, 16:])
        torch.set_num_threads(before)
        order = torch.from_numpy(np.asarray(batches[0].order, dtype=np.int64))  # resident batches are packed: launch row i = host row order[i]
        err = (out.logits_b.cpu() - ref["logits_b"][order]).abs()
        feat_err = float((out.features_be.cpu() - ref["features_be"][order]).abs().max())
        parity = {"read_sets": int(err.numel()), "max_logit_err": float(err.max()), "p9999_logit_err": float(torch.quantile(err, 0.9999)),
                  "median_logit_err": float(err.median()), "logit_errs_over_1e4": int((err > 1e-4).sum()), "max_feature_err": feat_err,
                  "contract": "BASELINE.json north_star: per-variant artifact logits within 1e-4 (fp32) of the reference CPU path; the run "
                              "fails above it (fp32 mode)"}
        note(f"parity: max |logit - oracle| over the {err.numel()} variants of resident batch 0 = {parity['max_logit_err']:.3e} "
             f"(99.99th percentile {parity['p9999_logit_err']:.3e})")
        finite = bool(torch.isfinite(out.logits_b).all() and torch.isfinite(out.features_be).all()
                      and torch.isfinite(eng.space.theta).all() and torch.isfinite(eng.space.gtheta).all())
        ok = parity["max_logit_err"] <= 1e-4
        parity["variants_needing_fp64_adjudication"] = parity["logit_errs_over_1e4"]
        parity["logit_errs_over_1e4_per_million"] = 1e6 * parity["logit_errs_over_1e4"] / max(1, parity["read_sets"])
        if finite and not ok and args.dtype == "f32" and not args.strict_parity:
            # The fp32 reference path is itself a few 1e-5 from the exact result on its worst variants (tests/test_scale_gpu.py
            # measured it over 2^20 variants).  A variant beyond 1e-4 is therefore recomputed by the oracle in fp64: the run
            # fails unless EVERY such variant's HIP result is within 1e-4 of the fp64 result and no further from it than 1.5 x
            # the fp32 oracle's own distance -- i.e. unless the excess is the reference's rounding, not the kernel's.
            from tests.helpers import oracle_forward, variant_rows
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.numel())
            bad_launch = torch.nonzero(err > 1e-4).view(-1)           # positions in the launch
            bad_host = order[bad_launch].numpy()                       # rows of the host arrays
            ref64 = oracle_forward(sd, config_for("p0"), ints[bad_host], floats[bad_host], variant_rows(ints, packed, bad_host), dtype=torch.float64)["logits_b"]
            hip_d = (out.logits_b.cpu()[bad_launch].double() - ref64).abs()
            o32_d = (ref["logits_b"][order][bad_launch].double() - ref64).abs()
            parity.update(max_hip_vs_fp64=float(hip_d.max()), max_fp32_oracle_vs_fp64=float(o32_d.max()))
            ok = bool(hip_d.max() <= 1e-4 and hip_d.max() <= 1.5 * max(float(o32_d.max()), 2e-5) and parity["logit_errs_over_1e4"] <= 4)
            parity["excess_is_the_fp32_reference_rounding"] = ok
            note(f"parity: {parity['logit_errs_over_1e4']} variant(s) beyond 1e-4 of the fp32 oracle; against fp64: HIP {hip_d.max():.3e}, fp32 oracle {o32_d.max():.3e}")
        if not finite or (args.dtype == "f32" and not ok):
            raise SystemExit(f"bench: outputs diverge from the oracle ({parity}, finite {finite})")

    if rank == 0:
        head = "train" if "train" in results else "filter"
        elapsed, roof, stats = results[head]
        value = world * args.batch * args.steps / elapsed
        line = {
            "metric": "read-sets/sec (train fwd+bwd)" if head == "train" else "read-sets/sec (filter fwd)",
            "value": value, "unit": "read-sets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, **stats, "preroll_s": prerolled.get(head, 0.0), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.dtype != "f32" else "f32 (fp32 accumulate; fwd: 2 x f16 pieces per operand = 23 bits; bwd: 2 x bf16 pieces = 16 bits)",
            "fwd_operand_bits": 23 if args.dtype == "f32" else 8, "bwd_operand_bits": 16 if args.dtype == "f32" else 8,
            "data": "synthetic" if args.data == "resident" else "synthetic, streamed through the device chunk loader (H2D inclusive)",
            "config": {"workload": ("train_model" if head == "train" else "filter_variants forward")
                       + f" on synthetic 1M-variant-scale {args.depth.upper()} ReadSet batches, hyperparameters P0 (59845 params)",
                       "batch_read_sets_per_gpu": args.batch, "mean_reads_per_set": reads_per_batch / args.batch,
                       "workgroups_per_batch": groups_per_batch,  # what a batch costs: rounds of 256 (backward) / 512 (forward)
                       "step": ("fwd + losses + bwd + bucketed grad all-reduce + clip + AdamW" if head == "train"
                                else "compute_batch_output under inference_mode"),
                       "parallelism": f"dp{world}" if world > 1 else "single", "kernels_sha": kernels_sha()},
            "roofline": roof,
        }
        if head == "train" and "filter" in results:
            ef, rf, sf = results["filter"]
            line["filter"] = {"metric": "read-sets/sec (filter fwd)", "value": world * args.batch * args.steps / ef,
                              "unit": "read-sets/s", "ms_per_step": 1e3 * ef / args.steps, **sf, "steps": args.steps,
                              "step": "compute_batch_output under inference_mode, same resident batches", "roofline": rf}
        if reduce_grads is not None:  # what RCCL saw: the driver can check that N ranks took part and the overlap hook fired
            line["collective"] = reduce_grads.describe()
        if sharded_filter is not None:
            line["filter_sharded_dataset"] = sharded_filter
        if small is not None:
            line["small_batch"] = small
        if stress is not None:
            line["stress"] = stress
        if loader is not None:
            line["loader"] = loader
            line["epoch"] = epoch
        if dropout is not None:
            line["dropout"] = dropout
        if six is not None:
            line["six_mfma_backward"] = six
        if shapes is not None:
            line["shapes"] = shapes
        if parity is not None:
            line["parity_check_max_logit_err"] = parity["max_logit_err"]
            line["parity_check"] = parity
            sweep = committed_parity_sweep()
            if sweep is not None:
                line["parity_sweep"] = sweep
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
