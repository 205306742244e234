#!/usr/bin/env python3
"""Throughput of the preshape hot path on MI355X: scenes/sec at 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one eval forward of ``ProxyTransformationNormReverse`` over this rank's batch of
synthetic scenes (BASELINE.json configs[1] shape: 100k points, 8^3 grid -> 256 kept clusters,
64 text + 196 image proxies, d = 256, image features stored as bf16 as that config names; 4 scenes
per GPU = the per-GPU shard of configs[2]).  All arithmetic is fp32; the rate with fp32-stored
features is measured in the same run and reported as `value_f32_features`.
Inputs are resident in HBM before the timed region; a step returns when the list of output
tensors exists (the host waits for the per-scene lengths, which the clustering chain publishes
early; the tensors' contents are stream-ordered like any torch result) and the timed region is
closed by a full device synchronise, so every step's GPU work is inside it.
Scenes are sharded by scene id with no data-path collective (weak scaling).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel (the one pass over img_feat after the means: k_img_pool for bf16-stored
                features, k_img_scores for fp32): algorithmic
                bytes per launch / average launch duration, timed with HIP events recorded by
                the library on the kernel's own stream during the timed steps; `traffic` = HBM
                bytes per launch from the committed rocprofv3 PMC pass (profiles/pmc_traffic.json,
                FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes, + WRITE_SIZE), null if the
                file has no entry for this kernel / shape
  cpu_baseline  the CPU oracle (torch CPU fp32 + C ball query / FPS, "port") on this box's
                host cores, same workload, bounded sample (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from proxytransformation_amd import MODELS, _abi                                   # noqa: E402
from proxytransformation_amd.synth import CONFIGS, fill_state_dict, make_scene_batch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--scenes-per-gpu", type=int, default=None)
    ap.add_argument("--img-dtype", default="bf16", choices=["f32", "bf16", "f16"],
                    help="storage type of the image features (BASELINE config 2 names bf16; arithmetic is fp32 "
                         "either way); the fp32-feature rate is reported next to it")
    ap.add_argument("--time-kernel", default="img_pass2",
                    help="launch site timed for the roofline object (img_pass2 = the dominant stream over img_feat "
                         "after the mean pass: k_img_pool for bf16 features, k_img_scores for fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=None, help="scenes in the CPU baseline sample")
    ap.add_argument("--breakdown", action="store_true", help="print per-kernel event timings to stderr")
    return ap.parse_args()


def build_module(cfg, device):
    mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
    sd = fill_state_dict(mod.state_dict())
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return mod.eval().to(device), sd


def kernel_id(lib, name):
    names = [lib.ptx_kernel_name(i).decode() for i in range(lib.ptx_kernel_count())]
    if name not in names:
        raise SystemExit(f"--time-kernel must be one of {names}")
    return names.index(name), names


def algorithmic_bytes(cfg, B, name, img_itemsize=4, dt="f32"):
    """Algorithmic HBM bytes one launch of `name` must move (kernel table)."""
    img = B * cfg.V * cfg.input_dim * cfg.img_spacial_dim ** 2 * img_itemsize
    table = {
        "k_img_mean": img, "img_pass2": img,
        "img_pass3": None if dt in ("bf16", "f16") else img,
        "k_minmax": B * cfg.N * 12,
        "k_affine<compact>": B * cfg.N * (12 + 4 + 12),
        "k_tile_count": B * cfg.N * 4,
    }
    return table.get(name)


# the kernel behind a launch site depends on the storage type of the image features
SITE_KERNEL = {
    "img_pass2": {"bf16": "k_img_pool", "f32": "k_img_scores", "f16": "k_img_pool"},
    "img_pass3": {"bf16": None, "f32": "k_img_gather", "f16": None},      # 16-bit features: no third launch
    "k_img_mean": {"bf16": "k_img_mean16", "f32": "k_img_mean", "f16": "k_img_mean16"},
}


def site_kernel(site, dt):
    return SITE_KERNEL.get(site, {}).get(dt, site)


def pmc_traffic(kernel, dt, cfg, B):
    """HBM bytes per launch of `kernel` from the committed PMC digest (tools/profile_round.sh)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if d.get("config") != cfg.name or d.get("scenes_per_gpu") != B:
            return None
        for name, v in d[dt].items():
            if kernel in name:
                return int(v["fetch_bytes"] + v["write_bytes"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(cfg, sd, n_scenes):
    """Time the CPU oracle (the checker, here only as the reported baseline) on a bounded sample.

    torch's CPU kernels scale badly past a few dozen threads on these short ops, so a small sweep
    of thread counts is timed (~4 s each) and the best one is reported with the threads it used."""
    from oracle import oracle
    ncpu = os.cpu_count() or 1
    pts, text, mask, img = make_scene_batch(cfg, scene_ids=range(n_scenes))
    kw = dict(grid_size=cfg.grid_size, dynamic_drop_radio=cfg.dynamic_drop_radio, num_sub=cfg.num_sub,
              num_heads=cfg.num_heads, text_blocks=cfg.text_blocks, img_blocks=cfg.img_blocks,
              points=pts, text_feats=text, text_mask=mask, img_feat=img)
    best = None
    tried = []
    for threads in sorted({min(ncpu, t) for t in (8, 32, 96)}):
        oracle.forward(sd, **kw, num_threads=threads)          # warm-up (lib load, thread pool)
        reps, t0 = 0, time.perf_counter()
        while True:
            oracle.forward(sd, **kw, num_threads=threads)
            reps += 1
            el = time.perf_counter() - t0
            if el > 4.0 or reps >= 10:
                break
        rate = n_scenes * reps / el
        tried.append(f"{threads}t: {rate:.2f}/s")
        if best is None or rate > best[0]:
            best = (rate, threads, reps, el)
    rate, threads, reps, el = best
    return dict(value=round(rate, 4), unit="scenes/s", cores=threads, kind="port",
                sample=f"{reps} forwards of {n_scenes} {cfg.name}-shape scenes in {el:.1f} s on {threads} of {ncpu} "
                       f"host threads (best of sweep {', '.join(tried)}); oracle/oracle.py = torch-CPU fp32 + "
                       f"single-thread C ball query / FPS")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU over RCCL, same flags
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = CONFIGS[args.config]
    B = args.scenes_per_gpu or cfg.B
    scene_ids = range(rank * B, rank * B + B)                  # weak scaling: B scenes per GPU
    mod, sd = build_module(cfg, device)
    pts, text, mask, img = make_scene_batch(cfg, scene_ids=scene_ids)
    points = [torch.from_numpy(p).to(device) for p in pts]
    text_dict = {"text_feats": torch.from_numpy(text).to(device),
                 "text_token_mask": torch.from_numpy(mask).to(device)}
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.img_dtype]
    img_f32 = torch.from_numpy(img).to(device)
    img_feat = img_f32 if tdt is torch.float32 else img_f32.to(tdt)
    lib = _abi.lib()
    kid, names = kernel_id(lib, args.time_kernel)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            outs = mod(points, text_dict, img_feat)
        n_out = sum(int(o.shape[0]) for o in outs) if args.warmup else None
        lib.ptx_timing_select(kid)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = mod(points, text_dict, img_feat)
        barrier()
        elapsed = time.perf_counter() - t0
        launches, total_ms = ctypes.c_int(0), ctypes.c_float(0.0)
        lib.ptx_timing_read(ctypes.byref(launches), ctypes.byref(total_ms))
        lib.ptx_timing_select(-1)

        # the same workload with fp32-stored image features (the reference's non-AMP layout)
        elapsed_f32 = None
        if tdt is not torch.float32:
            for _ in range(max(2, args.warmup // 2)):
                mod(points, text_dict, img_f32)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                mod(points, text_dict, img_f32)
            barrier()
            elapsed_f32 = time.perf_counter() - t1

        breakdown = None
        if args.breakdown and rank == 0:
            breakdown = {}
            for i, nm in enumerate(names):
                lib.ptx_timing_select(i)
                for _ in range(5):
                    mod(points, text_dict, img_feat)
                torch.cuda.synchronize()
                n_, ms_ = ctypes.c_int(0), ctypes.c_float(0.0)
                lib.ptx_timing_read(ctypes.byref(n_), ctypes.byref(ms_))
                breakdown[nm] = round(1e3 * ms_.value / max(n_.value, 1), 2)
            lib.ptx_timing_select(-1)
            print("per-kernel us/launch:", json.dumps(breakdown), file=sys.stderr)

    t = torch.tensor([elapsed, elapsed_f32 or 0.0], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    if elapsed_f32 is not None:
        elapsed_f32 = float(t[1].item())

    if rank == 0:
        total_scenes = world * B * args.steps
        abytes = algorithmic_bytes(cfg, B, args.time_kernel, img_feat.element_size(), args.img_dtype)
        roof = None
        if launches.value > 0 and not abytes:
            roof = dict(kernel=site_kernel(args.time_kernel, args.img_dtype), avg_launch_us=round(total_ms.value / launches.value * 1e3, 2))
        if launches.value > 0 and abytes:
            # the image branch is launched once per slice of scenes (2 slices per forward): the
            # algorithmic bytes of a step are spread over the launches actually recorded
            abytes = abytes * args.steps // launches.value
            avg_s = total_ms.value / launches.value / 1e3
            ach = abytes / avg_s / 1e9
            roof = dict(bound="hbm", kernel=site_kernel(args.time_kernel, args.img_dtype), achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=pmc_traffic(site_kernel(args.time_kernel, args.img_dtype), args.img_dtype, cfg, B),
                        avg_launch_us=round(avg_s * 1e6, 2), launches=launches.value,
                        algorithmic_bytes_per_launch=abytes)
        line = dict(metric="scenes/sec (100k pts, 256 clusters, 64 proxies)", value=round(total_scenes / elapsed, 2),
                    unit="scenes/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(1e3 * elapsed / args.steps, 4), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"{cfg.name}: N={cfg.N} pts, gs={cfg.grid_size}->M'={cfg.M_keep} kept clusters, "
                                         f"L={cfg.L} text + V={cfg.V} image proxies, d={cfg.embed_dim}, eval forward",
                                scenes_per_gpu=B, global_scenes_per_step=world * B, sharding="by scene, no collective",
                                img_feat_dtype=args.img_dtype, arithmetic="fp32 (exact-fp32 MFMA / VALU; 16-bit matrix pipe only through exact 3-way operand splits, fp32 accumulate)",
                                surviving_points_per_step=n_out),
                    roofline=roof)
        if elapsed_f32 is not None:
            line["value_f32_features"] = round(total_scenes / elapsed_f32, 2)
            line["ms_per_step_f32_features"] = round(1e3 * elapsed_f32 / args.steps, 4)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, sd, args.cpu_scenes or min(B, 2))
            except Exception as e:                               # the baseline must never sink the GPU number
                line["cpu_baseline"] = dict(value=None, error=repr(e))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
