#!/usr/bin/env python3
"""Benchmark of the Permutect artifact-model hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): read-sets/sec.  Default workload = BASELINE.json configs[1]: train_model on a synthetic
1M-variant WGS-shaped dataset with the production-shaped hyperparameters P0 (59 845 parameters): one "step" is
forward + losses + backward + (DP: RCCL gradient all-reduce) + global-norm clip + AdamW on one prepared batch that is
already resident in HBM.  `--mode filter` times the filter_variants forward (configs[2]) instead.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the timed region) and
`cpu_baseline` (the CPU oracle = PyTorch-CPU restatement of the reference path, timed on this box's host cores on a
bounded sample; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from permutect_amd.architecture.artifact_model import ArtifactModel  # noqa: E402
from permutect_amd.data.batch import Batch  # noqa: E402
from permutect_amd.parameters import P0_DIMS, p0_params  # noqa: E402
from permutect_amd.training.optimizer import FusedClipAdamW  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
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


def pmc_traffic(kernel, args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json:
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950, + WRITE_SIZE),
    or None when no committed measurement matches this workload."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("batch_read_sets") != args.batch or rec.get("depth") != args.depth:
        return None
    k = rec.get("kernels", {}).get(kernel)
    return None if k is None else k["hbm_bytes_per_launch"]


def algorithmic_macs_per_read(model, padded=False):
    """Forward MACs per read of the read-set path: read MLP + L gated blocks + reducer + rotation.
    `padded`: with every dimension rounded up to the 16-wide MFMA tile (what the matrix core actually executes)."""
    d = model.engine().plan.desc
    up = (lambda n: (n + 15) // 16 * 16) if padded else (lambda n: n)
    macs = 0
    used = set()

    def mlp(m):
        nonlocal macs
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


def cpu_baseline(mode, seconds=15.0):
    """The reference path as restated by oracle/artifact_oracle.py, on the host cores, bounded sample."""
    from oracle import artifact_oracle as O  # checker / baseline only -- never the product path
    from tests.helpers import config_for

    torch.manual_seed(0)
    cfg = config_for("p0")
    ref_model = ArtifactModel(p0_params(), device=torch.device("cpu"), **P0_DIMS)
    sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    b = 8192
    ints, floats, packed = synth_arrays(np.random.default_rng(1), b, "wgs")
    batch = dict(reads_re=torch.from_numpy(O.decode_packed_reads(packed).astype(np.float32)),
                 nref=torch.from_numpy(ints[:, 0].astype(np.int64)), nalt=torch.from_numpy(ints[:, 1].astype(np.int64)),
                 labels=torch.from_numpy(ints[:, 2].astype(np.int64)), sources=torch.zeros(b, dtype=torch.int64),
                 info_be=torch.from_numpy(floats[:, 6:].astype(np.float32)),
                 haplotypes_bh=torch.from_numpy(ints[:, 16:].astype(np.int64)))
    names = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(".base")]
    m = [torch.zeros_like(sd[n]) for n in names]
    v = [torch.zeros_like(sd[n]) for n in names]

    def step(i):
        if mode == "train":
            _, _, grads = O.train_step_grads(sd, cfg, batch)
            O.clip_and_adamw([sd[n] for n in names], [grads[n] for n in names], m, v, step=i + 1, lr=1e-3, weight_decay=0.01)
        else:
            with torch.inference_mode():
                O.compute_batch_output(sd, cfg, batch["reads_re"], batch["nref"], batch["nalt"], batch["info_be"], batch["haplotypes_bh"])

    step(0)
    t0, n = time.perf_counter(), 0
    while n < 3 or time.perf_counter() - t0 < seconds:
        step(n + 1)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": b * n / dt, "unit": "read-sets/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} {mode} steps of B={b} WGS-shaped read sets ({packed.shape[0]} reads), P0, fp32, "
                      f"oracle/artifact_oracle.py (PyTorch-CPU restatement of the reference path)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "filter"], default="train")
    ap.add_argument("--batch", type=int, default=65536, help="read sets per step per GPU")
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct synthetic batches kept in HBM per GPU")
    ap.add_argument("--depth", choices=["wgs", "high", "stress"], default="wgs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
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
    model.train(args.mode == "train")
    opt = FusedClipAdamW(model, lr=1e-3, weight_decay=0.01)
    eng = model.engine()

    rng = np.random.default_rng(1000 + rank)  # each rank owns a different shard of the synthetic dataset
    batches, reads_total = [], 0
    stream_batches = None
    if args.data == "resident":
        for _ in range(args.resident_batches):
            ints, floats, packed = synth_arrays(rng, args.batch, args.depth)
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

        def endless():
            while True:
                # (training draws shuffled batches; filter_variants walks the candidates in order)
                for cb in dataset.device_loader(args.batch, dev, chunk_variants=args.chunk_variants, rng=rng, shuffle=args.mode == "train"):
                    if cb.size() == args.batch:
                        yield cb
        stream_batches = endless()
    torch.cuda.synchronize()

    def all_reduce_grads(flat):
        if dist is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)  # loss is a batch SUM (reference artifact_model.py:90)

    def step(i):
        batch = batches[i % len(batches)] if stream_batches is None else next(stream_batches)
        if args.mode == "train":
            opt.zero_grad()
            out = model.compute_batch_output(batch)
            losses = model.compute_batch_losses(out, batch)
            losses.total_loss.backward()
            opt.step(pre_reduce=all_reduce_grads)
        else:
            with torch.inference_mode():
                model.compute_batch_output(batch)

    for i in range(args.warmup):
        step(i)
    eng.timers = {"pmt_forward": [], "pmt_backward": []}
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = {k: (sum(s.elapsed_time(e) for s, e in v) / len(v) if v else None) for k, v in eng.timers.items()}
    eng.timers = None
    if rank == 0:
        macs = algorithmic_macs_per_read(model)
        fwd_flops = 2.0 * macs * reads_per_batch
        if args.mode == "train":
            dom, dom_flops = "pmt_backward_kernel", 2.0 * fwd_flops  # dgrad + wgrad; the in-kernel recompute is not counted
            dom_ms = kernel_ms["pmt_backward"]
        else:
            dom, dom_flops, dom_ms = "pmt_forward_kernel", fwd_flops, kernel_ms["pmt_forward"]
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        traffic = pmc_traffic(dom, args)
        pad_ratio = algorithmic_macs_per_read(model, padded=True) / macs
        value = world * args.batch * args.steps / elapsed
        line = {
            "metric": "read-sets/sec (train fwd+bwd)" if args.mode == "train" else "read-sets/sec (filter fwd)",
            "value": value, "unit": "read-sets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if args.data == "resident" else "synthetic, streamed through the device chunk loader (H2D inclusive)",
            "config": {"workload": ("train_model" if args.mode == "train" else "filter_variants forward")
                       + f" on synthetic 1M-variant-scale {args.depth.upper()} ReadSet batches, hyperparameters P0 (59845 params)",
                       "batch_read_sets_per_gpu": args.batch, "mean_reads_per_set": reads_per_batch / args.batch,
                       "workgroups_per_batch": groups_per_batch,  # what a batch costs: rounds of 256 (backward) / 512 (forward)
                       "step": "fwd + losses + bwd + grad all-reduce + clip + AdamW" if args.mode == "train" else "compute_batch_output under inference_mode",
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "kernel": dom,
                         "kernel_ms": dom_ms, "algorithmic_flops_per_launch": dom_flops,
                         "matrix_pipe": "fp32-equivalent: forward / recompute / dgrad as six bf16 MFMAs on three-piece splits of both "
                                        "operands, wgrad as exact fp32 MFMAs; `peak` is the dense fp32 MFMA rate",
                         "padded_flops_per_launch": dom_flops * pad_ratio, "frac_padded": achieved * pad_ratio / PEAK_FP32_MFMA_TFLOPS,
                         "hbm_frac": None if traffic is None else traffic / (dom_ms * 1e-3) / (PEAK_HBM_GBS * 1e9),
                         "other_kernel_ms": {k: v for k, v in kernel_ms.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.mode)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
