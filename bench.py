"""Benchmark of the hot path: the CycleGAN training step of the reference (enhanced_train.py:59-131) -- 6 generator
forwards + 10 discriminator forwards, backward through all of them, two Adam steps -- on N MI355X, one process per
GPU, image batches sharded data-parallel with one RCCL all-reduce per optimizer.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one train step on a per-rank batch of 32 image pairs at 256x256 (synthetic uniform [-1,1) data, random
init of the reference architecture at channels=16: what enhanced_train.py:18-21 trains).  Weak scaling: the per-GPU
batch is fixed.  Rank 0 prints ONE JSON line.  An "image" is one input image pushed through the step; a step consumes
2 * batch images (one per domain), so value = 2 * batch * N / step_time.

Besides the contract fields the line carries
  roofline     : the kernel symbol with the largest share of GPU time, measured with HIP events around every launch of
                 an instrumented step that follows the timed region (same stream, same shapes): achieved = sum of
                 algorithmic FLOPs (or bytes) of its launches / sum of their durations, against the MI355X peak;
  cpu_baseline : the oracle's restatement of the same train step (oracle/restatement.py, validated against the reference
                 in the build container) timed on this box's host cores on a bounded sample (batch 1, 256x256).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multi-style-transfer-gan_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0         # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--channels", type=int, default=16)
    ap.add_argument("--style-loss", action="store_true",
                    help="add the build-defined VGG/Gram multi-style loss (3 weighted references) to the step; NOT the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel table of the instrumented step to stderr")
    return ap.parse_args()


def cpu_baseline(channels: int, size: int):
    """The oracle's train step on the host cores: batch 1, 1 warm-up + timed steps bounded to ~20 s."""
    from oracle import restatement as R
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    sds = [R.make_state_dict(R.generator_spec(channels), 11), R.make_state_dict(R.generator_spec(channels), 12),
           R.make_state_dict(R.discriminator_spec(channels), 13), R.make_state_dict(R.discriminator_spec(channels), 14)]
    model = R.CycleGANOracle(*sds)
    a, b = R.make_input((1, 3, size, size), 21), R.make_input((1, 3, size, size), 22)
    model.train_step(a, b)
    t0, n = time.perf_counter(), 0
    while n < 3 or (time.perf_counter() - t0 < 12.0 and n < 8):
        model.train_step(a, b)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2.0 / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle train step, batch 1 pair at {size}x{size}, channels={channels}, {n} timed steps after 1 warm-up, "
                      f"{threads} torch threads, fp32"}


def pmc_traffic(sym: str, args):
    """HBM bytes per launch of `sym` from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs,
    read side doubled as the MI355X guide prescribes for gfx950; tools/gpu_refresh.sh + tools/pmc_traffic.py).  PMC
    counters cannot be read from inside this process, so the figure is the one measured on the default workload and is
    reported only when this run IS the default workload."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    default = args.batch == 32 and args.size == 256 and args.channels == 16 and not args.style_loss
    if not default or not os.path.exists(path):
        return None, None
    t = json.load(open(path)).get(sym)
    return (t["hbm_bytes"], "profiles/r01_pmc_traffic.json") if t else (None, None)


def main():
    args = parse()
    from mstg_hip import dp, ops
    # MSTG_BENCH_REHEARSE=1: rehearsal of the N > 1 flow on a one-GPU box -- gloo instead of RCCL, every rank on cuda:0
    rehearse = os.environ.get("MSTG_BENCH_REHEARSE", "0") == "1"
    local = dp.init_from_env("gloo" if rehearse else "nccl")
    if rehearse:
        local = 0
    world = dp.world_size()
    if world != args.gpus:
        if dp.rank() == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1", file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import enhanced_train
    torch.manual_seed(42)  # reference seeds with set_seed(42) (pretrain.py:13-17); every rank builds the same weights
    model = enhanced_train.EnhancedCycleGAN(channels=args.channels, num_transformer_blocks=0, device=dev)
    model.sync_replicas()
    if args.style_loss:
        sgen = [torch.Generator().manual_seed(2001 + k) for k in range(3)]
        refs = [(torch.rand((4, 3, args.size, args.size), generator=g_) * 2 - 1) for g_ in sgen]
        model.attach_style_loss(refs, (0.5, 0.3, 0.2), lambda_style=1.0)
    gen = torch.Generator().manual_seed(1234 + dp.rank())
    shape = (args.batch, 3, args.size, args.size)
    real_A = (torch.rand(shape, generator=gen) * 2 - 1).to(dev)
    real_B = (torch.rand(shape, generator=gen) * 2 - 1).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.train_step_async(real_A, real_B)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = model.train_step_async(real_A, real_B)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(torch.isfinite(losses).all())
    ms = dt / args.steps * 1e3
    value = 2.0 * args.batch * world / (dt / args.steps)

    roofline = None
    if not args.no_roofline:
        # EVERY rank runs the instrumented step (it contains the two gradient all-reduces); only rank 0 reports
        ops.KernelTimer.enabled, ops.KernelTimer.records = True, []
        model.train_step_async(real_A, real_B)
        torch.cuda.synchronize()
        ops.KernelTimer.enabled = False
    if not args.no_roofline and dp.rank() == 0:
        table = ops.KernelTimer.summary()
        total_ms = sum(r["ms"] for r in table.values())
        if args.kernel_table:
            for sym, r in sorted(ops.KernelTimer.summary(detail=True).items(), key=lambda kv: -kv[1]["ms"])[:200]:
                print(f"[kernels] {sym:72s} launches {r['launches']:5d}  {r['ms']:9.3f} ms  {100 * r['ms'] / total_ms:5.1f}%  "
                      f"{r['flops'] / r['ms'] / 1e9 if r['ms'] else 0:8.2f} TFLOP/s  {r['bytes'] / r['ms'] / 1e6 if r['ms'] else 0:9.1f} GB/s",
                      file=sys.stderr)
        # the timer brackets C-ABI calls; a call that launches two kernels (norm_act_fwd / norm_act_bwd = partial + apply) cannot
        # be attributed per kernel, so the dominant KERNEL is taken among the single-kernel calls (their names are the symbols
        # rocprofv3 reports); the two-kernel calls stay in --kernel-table
        sym, r = max(((k, v) for k, v in table.items() if "_kernel" in k), key=lambda kv: kv[1]["ms"])
        sec = r["ms"] / 1e3
        t_mfma, t_hbm = r["flops"] / (PEAK_F32_MFMA_TFLOPS * 1e12), r["bytes"] / (PEAK_HBM_GBS * 1e9)
        if t_mfma >= t_hbm:
            ach = r["flops"] / sec / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None}
        else:
            ach = r["bytes"] / sec / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
        roofline["algorithmic_bytes_per_launch"] = round(r["bytes"] / r["launches"])
        roofline["traffic"], roofline["traffic_source"] = pmc_traffic(sym, args)
        # per-launch durations are taken with the step on ONE stream (EnhancedCycleGAN drops its two side streams while
        # ops.KernelTimer is enabled): on two streams launches overlap and a launch's event-to-event time includes its neighbour
        roofline["measured"] = "instrumented step on one stream (MSTG_STREAMS=0 equivalent); timed region runs on two"
        roofline.update({"kernel": sym, "launches_per_step": r["launches"], "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2),
                         "share_of_gpu_time": round(r["ms"] / total_ms, 3), "instrumented_step_gpu_ms": round(total_ms, 2)})

    cpu = None
    if dp.rank() == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.channels, args.size)

    if dp.rank() == 0:
        line = {"metric": "train-step images/sec (fwd+bwd+loss), 256x256 batch32", "value": round(value, 2), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"reference CycleGAN train_step (2x EnhancedGenerator + 2x EnhancedDiscriminator, channels={args.channels}, "
                                       f"num_transformer_blocks=0), {args.batch} image pairs/GPU at {args.size}x{args.size}, fwd+bwd+loss+Adam"
                                       + (" + build-defined VGG16/Gram multi-style loss (3 refs)" if args.style_loss else ""),
                           "pairs_per_gpu": args.batch, "images_per_step_per_gpu": 2 * args.batch, "size": args.size,
                           "channels": args.channels, "parallelism": f"dp{world}", "losses_finite": finite},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
