#!/usr/bin/env python3
"""Throughput of the preshape hot path on MI355X: scenes/sec at 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W        (N > 1: re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one eval forward of ``ProxyTransformationNormReverse`` over this rank's batch of synthetic scenes
(BASELINE.json configs[1] shape: 100k points, 8^3 grid -> 256 kept clusters, 64 text + 196 image proxies, d = 256,
image features stored as bf16 as that config names; 4 scenes per GPU = the per-GPU shard of configs[2]).  All
arithmetic is fp32.  Inputs are resident in HBM before the timed region; consecutive steps use DIFFERENT input
sets (``--sets``, default 3: together larger than the 256 MiB Infinity Cache), so every step streams its image
features from HBM.  A step returns when the list of output tensors exists (the host waits for the per-scene
lengths, which the clustering chain publishes early; the tensors' contents are stream-ordered like any torch result)
and the timed region is closed by a full device synchronise, so every step's GPU work is inside it.
Scenes are sharded by scene id with no data-path collective (weak scaling).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline         dominant kernel (the pass over img_feat after the means: k_img_pool for 16-bit features, k_img_scores
                   for fp32): algorithmic bytes per launch / average launch duration, timed with HIP events recorded by
                   the library on the kernel's own stream during the timed steps; `traffic` = HBM bytes per launch from
                   the committed rocprofv3 PMC passes (profiles/r02_pmc_traffic.json: FETCH_SIZE corrected as
                   MI355X_MICROARCH.md prescribes, + WRITE_SIZE), null if that file has no entry for this shape
  roofline_passes  per-pass figures of the north star, measured after the timed region with every launch site
                   bracketed by events (which perturbs the step a little -- hence not inside it), at the benchmark's
                   4 scenes per GPU and at 32 scenes per GPU: HBM fraction of the clustering / apply passes, fp32-MFMA
                   fraction of proxy attention and of the block GEMMs
  cpu_baseline     the CPU oracle (torch CPU fp32 + C ball query / FPS, "port") on this box's host cores, same
                   workload, bounded sample (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from proxytransformation_amd import MODELS, _abi                                   # noqa: E402
from proxytransformation_amd.synth import CONFIGS, fill_state_dict, make_scene_batch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix (v_mfma_f32_32x32x2_f32), 256 CUs
# the ceiling of the pipe the fp32-EQUIVALENT kernels really run on: k_gemm64x / k_gemm128x / k_proxy_attn / k_mlp
# issue six v_mfma_f32_32x32x16_bf16 per fp32-equivalent product (csrc/split3.h), so their ceiling is the dense bf16 peak / 6
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_SPLIT_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def so_sha16():
    """sha256 (first 16 hex digits) of the HIP library this process loaded: ties PMC digests to the binary they measured."""
    import hashlib
    try:
        return hashlib.sha256(open(_abi.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--scenes-per-gpu", type=int, default=None)
    ap.add_argument("--img-dtype", default=None, choices=["f32", "bf16", "f16"],
                    help="storage type of the image features (cfg2: bf16 as BASELINE names it, cfg5: fp16, otherwise "
                         "fp32; arithmetic is fp32 either way)")
    ap.add_argument("--sets", type=int, default=3, help="distinct input sets rotated through the steps")
    ap.add_argument("--streams", type=int, default=1,
                    help="lanes: torch streams the timed steps are issued on, one host thread per stream (a serving loop with "
                         "several forwards in flight).  Measured SLOWER than one lane at every count (profiles/r04_lanes_pipelining.txt): "
                         "the pooling pass needs every CU's register file, the neighbouring forward's small kernels take its slots")
    ap.add_argument("--single-thread", action="store_true", help="with --streams > 1: issue round-robin from ONE host thread")
    ap.add_argument("--time-kernel", default="img_pass2", help="launch site timed inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-passes", action="store_true", help="skip roofline_passes / fp32-feature extras")
    ap.add_argument("--cpu-scenes", type=int, default=None, help="scenes in the CPU baseline sample")
    ap.add_argument("--breakdown", action="store_true", help="print per-kernel event timings to stderr")
    ap.add_argument("--setup-forwards", type=int, default=32,
                    help="forwards of the untimed set-up phase in front of the W warm-up steps (parameter tables, workspace, pinned "
                         "buffers, every input set paged in -- and the device out of its idle state: with 6 of them and the "
                         "driver's --steps 20 --warmup 5 the 5 ms that are timed start 3 ms after the first kernel and read 4 %% "
                         "low, 15.7k vs 16.3-16.5k scenes/s; reported in config.setup_forwards)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed blocks of exactly --steps steps each (every block bracketed by barrier + synchronise, MAX over "
                         "ranks per block); `value` is the MEDIAN block, the spread is reported beside it (timed_blocks)")
    ap.add_argument("--time-every", type=int, default=4,
                    help="the roofline kernel is bracketed by HIP events on every n-th measured step (each event record costs "
                         "the stream ~6 us of idle between two kernels; 1 = every step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL)")
    ap.add_argument("--gemm-policy", type=int, default=None,
                    help="ptx_gemm_policy: 128x128 tiles in a launch from which the 128x128-tile GEMM is used (0 = never, 1 = whenever legal; default: the library's 256)")
    ap.add_argument("--pipeline", action="store_true",
                    help="BASELINE configs[3] as a pipeline: depth maps -> ingest -> neck -> 1 cm voxels -> four levels of image-feature "
                         "sampling, chained on one stream (proxytransformation_amd/pipeline.py); use with --config cfg4_room")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid for 1-GPU boxes: every rank uses cuda:0 (with --backend gloo); numbers are meaningless")
    return ap.parse_args()


def build_module(cfg, device):
    mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
    sd = fill_state_dict(mod.state_dict())
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return mod.eval().to(device), sd


# the kernel behind a launch site depends on the storage type of the image features
SITE_KERNEL = {
    "img_pass2": {"bf16": "k_img_pool", "f32": "k_img_pool32", "f16": "k_img_pool"},
    "img_pass3": {"bf16": None, "f32": None, "f16": None},      # r05: no third launch for fp32 features either (k_img_pool32)
    "k_img_mean": {"bf16": "k_img_mean16", "f32": "k_img_mean", "f16": "k_img_mean16"},
}


def site_kernel(site, dt):
    return SITE_KERNEL.get(site, {}).get(dt, site)


def pmc_traffic(kernel, dt, cfg, B):
    """(HBM bytes per launch of `kernel` from the committed PMC digest of tools/profile_round.sh, stale flag): stale =
    the digest was taken with another build of libproxyt_hip.so than the one loaded now."""
    try:
        d = json.load(open(PMC_FILE))
        if d.get("config") != cfg.name or d.get("scenes_per_gpu") != B:
            return None, None
        stale = d.get("so_sha16") != so_sha16()
        for name, v in d[dt].items():
            if kernel in name:
                return int(v["fetch_bytes"] + v["write_bytes"]), stale
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def work_model(cfg, B, dt_bytes):
    """Algorithmic bytes / flops of ONE launch of each site (DESIGN.md kernel table)."""
    N, M, K, Mk, C, L, V, H = cfg.N, cfg.M, cfg.num_sub, cfg.M_keep, cfg.embed_dim, cfg.L, cfg.V, cfg.embed_dim * 4
    R = B * Mk
    img = B * V * cfg.input_dim * cfg.img_spacial_dim ** 2 * dt_bytes
    byts = {
        "k_img_mean": img, "img_pass2": img, "img_pass3": None,
        "k_minmax": B * N * 12,
        # two ball-query passes (upper bound: every point read once per pass) + cluster writes of the second
        "k_cluster": B * (2 * 12 * N + M * K * (4 + 12) + M * 16),
        "k_tags": B * N * 4,
        "k_affine<compact>": B * N * (12 + 4 + 12),
    }
    flops = {
        "k_attn32[proxy_as_query]": 4.0 * R * (L + V) * C, "k_attn32[proxy_as_key]": 4.0 * R * (L + V) * C,
        "k_proxy_attn[fused]": 8.0 * R * (L + V) * C,            # both contractions in one launch (csrc/fattn.hip)
        "k_gemm_nt[qkv+proxy_proj]": 2.0 * 2 * R * 3 * C * C + 2.0 * B * L * C * C,
        "k_gemm_nt[pp_img]": 2.0 * B * V * C * C,
        "k_gemm_nt[proj]": 2.0 * 2 * R * C * C, "k_gemm_nt[fc1]": 2.0 * 2 * R * H * C, "k_gemm_nt[fc2]": 2.0 * 2 * R * H * C,
        "k_mlp[fc1+gelu+fc2]": 2.0 * 2 * 2 * R * H * C,
    }
    return byts, flops


def train_step_ms(device, steps=40):
    """One training step (train-mode forward + backward through every live parameter, text_feats and img_feat) at the
    reference's training shape -- CFG:41, 108, 145: 6 scenes per GPU, 100k points, gs = 12, 3 + 3 blocks, 20 views."""
    from proxytransformation_amd.synth import PreshapeConfig
    cfg = PreshapeConfig("cfg4train", B=6, N=100000, grid_size=12, dynamic_drop_radio=0.6, L=20, V=20, text_blocks=3,
                         img_blocks=3, seed_base=4500)
    mod = MODELS.build(dict(type="ProxyTransformationNormReverse", **cfg.module_kwargs()))
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(mod.state_dict()).items()})
    mod = mod.to(device).train()
    pts, text, mask, img = make_scene_batch(cfg)
    args = ([torch.from_numpy(p).to(device) for p in pts],
            {"text_feats": torch.from_numpy(text).to(device).requires_grad_(True), "text_token_mask": torch.from_numpy(mask).to(device)},
            torch.from_numpy(img).to(device).requires_grad_(True))

    leaves = list(mod.parameters()) + [args[1]["text_feats"], args[2]]

    gos = {}

    def step(scalar_loss=False):
        for t in leaves:                # optimizer.zero_grad(set_to_none=True): gradients are written, not accumulated
            t.grad = None
        outs = mod(*args)
        if scalar_loss:                 # a loss built from the outputs: six reductions + their backward on top of the neck's own work
            sum(o.sum() for o in outs).backward()
            return
        # the neck sits in the middle of the detector: the gradients of its outputs ARRIVE from the stages behind it
        key = tuple(o.shape[0] for o in outs)
        if key not in gos:
            gos[key] = [torch.ones_like(o) for o in outs]
        torch.autograd.backward(outs, gos[key])

    blocks = {}

    def timed(scalar_loss, label=None):
        # median of three blocks of `steps` steps: the step runs at the pace of the box's host whenever that is slower than the GPU,
        # and that pace moves by 30 % within seconds on a shared box (profiles/r05_train_host_pace.txt)
        for _ in range(10):             # the allocator's per-stream pools settle within the first few steps
            step(scalar_loss)
        vals = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(scalar_loss)
            torch.cuda.synchronize()
            vals.append(round(1e3 * (time.perf_counter() - t0) / steps, 3))
        if label:
            blocks[label] = vals
        return sorted(vals)[1]
    ms = timed(False, "ms")
    ms_loss = timed(True)
    mod.compute_dtype = "bf16"          # opt-in: the blocks' Linear layers and their gradients on plain bf16 operands (what --amp gives them)
    ms_bf16 = timed(False)
    mod.compute_dtype = "fp32"
    # the same step (forward + backward, drop rates 0) through the CPU oracle on this box's host cores: a bounded sample, the
    # checker used only as the reported baseline
    cpu = None
    try:
        from oracle import oracle
        threads = min(os.cpu_count() or 1, 32)
        sd0 = fill_state_dict(mod.state_dict())
        kw = dict(grid_size=cfg.grid_size, dynamic_drop_radio=cfg.dynamic_drop_radio, num_sub=cfg.num_sub, num_heads=cfg.num_heads,
                  text_blocks=cfg.text_blocks, img_blocks=cfg.img_blocks, points=pts, text_feats=text, text_mask=mask, img_feat=img,
                  num_threads=threads, timing_only=True)
        oracle.forward_train(sd0, **kw)                    # warm-up (library load, thread pool)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 and (reps == 0 or time.perf_counter() - t0 < 10.0):
            oracle.forward_train(sd0, **kw)
            reps += 1
        el = (time.perf_counter() - t0) / reps
        cpu = dict(value=round(1e3 * el, 1), unit="ms per step", cores=threads, kind="port",
                   sample=f"{reps} steps of oracle.forward_train (torch-CPU autograd + C ball query / FPS, drop rates 0) on {threads} of "
                          f"{os.cpu_count()} host threads")
    except Exception as e:                                 # never sinks the GPU numbers
        cpu = dict(value=None, error=repr(e))
    return dict(ms=ms, ms_blocks=blocks.get("ms"), ms_with_scalar_loss=ms_loss, ms_bf16_compute=ms_bf16, steps=steps, cpu_baseline=cpu,
                shape="6 scenes x 100k points, gs=12 -> 691 kept clusters, L=20, V=20 fp32 features, 3+3 blocks (CFG:41,108,145); "
                      "drop rates 0.2; forward + backward, output gradients handed in (ms) / a scalar loss built from the "
                      "outputs (ms_with_scalar_loss)")


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


class InputSets:
    """`nsets` device-resident input sets of B scenes each; set j of rank r holds scenes (j * world + r) * B ..."""

    def __init__(self, cfg, B, nsets, rank, world, device, tdt):
        self.sets = []
        for j in range(nsets):
            first = (j * world + rank) * B
            pts, text, mask, img = make_scene_batch(cfg, scene_ids=range(first, first + B))
            img_t = torch.from_numpy(img).to(device)
            self.sets.append(dict(
                points=[torch.from_numpy(p).to(device) for p in pts],
                text={"text_feats": torch.from_numpy(text).to(device), "text_token_mask": torch.from_numpy(mask).to(device)},
                img_f32=img_t, img=img_t if tdt is torch.float32 else img_t.to(tdt)))
            del img
        self.B = B

    def args(self, i, f32=False):
        s = self.sets[i % len(self.sets)]
        return s["points"], s["text"], (s["img_f32"] if f32 else s["img"])

    def widened(self, factor):
        """B * factor scenes per set out of the resident ones (distinct addresses: copies, not views)."""
        out = InputSets.__new__(InputSets)
        out.B = self.B * factor
        out.sets = []
        n = len(self.sets)
        for j in range(n):
            src = [self.sets[(j + k) % n] for k in range(factor)]
            out.sets.append(dict(
                points=[p.clone() for s in src for p in s["points"]],
                text={"text_feats": torch.cat([s["text"]["text_feats"] for s in src]),
                      "text_token_mask": torch.cat([s["text"]["text_token_mask"] for s in src])},
                img=torch.cat([s["img"] for s in src]), img_f32=None))
        return out


def timed_steps(mod, inputs, steps, barrier, streams=None, f32=False, start=0, workers=None):
    barrier()
    t0 = time.perf_counter()
    if workers is not None:
        outs = workers.run(start, steps, f32)
    elif streams is None:
        for i in range(start, start + steps):
            outs = mod(*inputs.args(i, f32))
    else:
        for i in range(start, start + steps):
            with torch.cuda.stream(streams[i % len(streams)]):
                outs = mod(*inputs.args(i, f32))
    barrier()
    return time.perf_counter() - t0, outs


class LaneWorkers:
    """One host thread per torch stream (lane), as a serving loop would drive the module: thread j issues forwards j, j + S,
    j + 2 S, ... of a block on its own stream (ctypes releases the GIL inside the library, so the lanes' enqueue and count waits
    overlap).  The threads are parked between blocks; a block is released and collected through barriers."""

    def __init__(self, mod, inputs, streams):
        import threading
        self.mod, self.inputs, self.streams = mod, inputs, streams
        self.start, self.done = threading.Barrier(len(streams) + 1), threading.Barrier(len(streams) + 1)
        self.job, self.outs, self.err, self.stop = None, [None] * len(streams), None, False
        self.threads = [threading.Thread(target=self._run, args=(j,), daemon=True) for j in range(len(streams))]
        for t in self.threads:
            t.start()

    def _run(self, j):
        with torch.no_grad():
            while True:
                self.start.wait()
                if self.stop:
                    return
                first, steps, f32 = self.job
                try:
                    with torch.cuda.stream(self.streams[j]):
                        for i in range(first + j, first + steps, len(self.streams)):
                            self.outs[j] = self.mod(*self.inputs.args(i, f32))
                except Exception as e:          # surfaced by run()
                    self.err = e
                self.done.wait()

    def run(self, first, steps, f32=False):
        self.job = (first, steps, f32)
        self.start.wait()
        self.done.wait()
        if self.err is not None:
            raise self.err
        return self.outs[(steps - 1) % len(self.streams)]

    def close(self):
        self.stop = True
        self.start.wait()


def site_times(lib, names, mod, inputs, steps):
    """Average duration of every launch site over `steps` forwards (all sites bracketed by events)."""
    lib.ptx_timing_select_mask((1 << len(names)) - 1)
    for i in range(steps):
        mod(*inputs.args(i))
    torch.cuda.synchronize()
    n = (ctypes.c_int * len(names))()
    ms = (ctypes.c_float * len(names))()
    lib.ptx_timing_read_sites(n, ms, len(names))
    lib.ptx_timing_select(-1)
    return {nm: 1e3 * ms[i] / n[i] for i, nm in enumerate(names) if n[i] > 0}


def prefix_max(mod, inputs):
    """P_max of SURVEY 8d's prefix-aware bound: the largest point index the (second) ball query scans in any scene of input set 0 --
    the early-exit query reads a PREFIX of the cloud per centre (first K hits in index order), not the scene."""
    d = mod.forward_debug(*inputs.args(0))
    torch.cuda.synchronize()
    return int(d["idx2"].max()) + 1


def passes_report(cfg, B, us, dt_bytes, p_max=None, step_s=None):
    byts, flops = work_model(cfg, B, dt_bytes)
    rep = {"scenes_per_gpu": B}

    def hbm(sites):
        t = sum(us[s] for s in sites if s in us)
        b = sum(byts[s] for s in sites if byts.get(s) and s in us)
        return dict(us=round(t, 2), algorithmic_MB=round(b / 1e6, 2), achieved_GBs=round(b / t / 1e3, 1),
                    frac_of_hbm_peak=round(b / t / 1e3 / HBM_PEAK_GBS, 4)) if t > 0 else None

    def mfma(sites):
        t = sum(us[s] for s in sites if s in us)
        f = sum(flops[s] for s in sites if s in us)
        return dict(us=round(t, 2), GFLOP=round(f / 1e9, 3), achieved_TFLOPs=round(f / t / 1e6, 2),
                    frac_of_f32_mfma_peak=round(f / t / 1e6 / MFMA_F32_PEAK_TFLOPS, 4),
                    frac_of_bf16_split_peak=round(f / t / 1e6 / MFMA_SPLIT_PEAK_TFLOPS, 4),
                    ceilings="fp32-equivalent FLOPs; f32 peak 157.3 TF = the fp32 matrix instruction's; split peak 416.7 TF = 2.5 PF bf16 / 6 "
                             "(the pipe these kernels use: six bf16 MFMAs per product, csrc/split3.h)") if t > 0 else None
    rep["clustering_pass_hbm"] = hbm(["k_minmax", "k_cluster"])
    rep["clustering_pass_hbm"]["bound"] = "upper: each ball-query pass reads every point once (SURVEY 8d: 12 N + 2 x 12 N + writes)"
    if p_max is not None and "k_minmax" in us and "k_cluster" in us:
        # SURVEY 8d's prefix-aware LOWER bound, reported alongside: the min / max pass reads the scene (12 N), each query pass only
        # the prefix it scans (12 P_max), plus the cluster writes.  This is the bound that matches the algorithm -- the early exit
        # is why the pass is latency-bound (one wave per centre walking ~P_max points), not a bandwidth kernel
        N, M, K = cfg.N, cfg.M, cfg.num_sub
        lo = B * (12 * N + 2 * 12 * p_max + M * K * (4 + 12) + M * 16)
        t = us["k_minmax"] + us["k_cluster"]
        rep["clustering_pass_hbm_prefix_bound"] = dict(
            us=round(t, 2), P_max=p_max, algorithmic_MB=round(lo / 1e6, 2), achieved_GBs=round(lo / t / 1e3, 1),
            frac_of_hbm_peak=round(lo / t / 1e3 / HBM_PEAK_GBS, 4),
            k_minmax_alone=dict(us=round(us["k_minmax"], 2), algorithmic_MB=round(B * 12 * N / 1e6, 2),
                                frac_of_hbm_peak=round(B * 12 * N / us["k_minmax"] / 1e3 / HBM_PEAK_GBS, 4)),
            what="12 N (min/max) + 2 x 12 P_max (the two early-exit query passes, P_max = largest scanned index of input set 0) + "
                 "cluster writes; k_minmax is the only kernel of the pass that streams the scene")
    rep["apply_pass_hbm"] = hbm(["k_tags", "k_affine<compact>"])
    rep["k_minmax"] = hbm(["k_minmax"])
    rep["k_affine"] = hbm(["k_affine<compact>"])
    rep["img_mean_pass_hbm"] = hbm(["k_img_mean"])
    rep["img_pool_pass_hbm"] = hbm(["img_pass2"])
    rep["img_features_hbm"] = hbm(["k_img_mean", "img_pass2", "img_pass3"])       # every streaming pass over img_feat together
    rep["proxy_attention_mfma"] = mfma(["k_attn32[proxy_as_query]", "k_attn32[proxy_as_key]", "k_proxy_attn[fused]"])
    rep["block_gemms_mfma"] = mfma(["k_gemm_nt[qkv+proxy_proj]", "k_gemm_nt[pp_img]", "k_gemm_nt[proj]",
                                    "k_gemm_nt[fc1]", "k_gemm_nt[fc2]", "k_mlp[fc1+gelu+fc2]"])
    if step_s is not None:
        # the whole step against the HBM roofline: SURVEY 8d's algorithmic bytes of one forward -- point side 60 N + 40 M K per scene
        # (min/max read, two query passes at their upper bound, cluster writes, apply read + write) + ONE read of img_feat -- over
        # the measured step time.  (The step is a dependency chain of ~10 launches at this batch: DESIGN.md 5.2.)
        N, M, K = cfg.N, cfg.M, cfg.num_sub
        img = B * cfg.V * cfg.input_dim * cfg.img_spacial_dim ** 2 * dt_bytes
        alg = B * (60 * N + 40 * M * K) + img
        rep["whole_step"] = dict(ms=round(1e3 * step_s, 4), algorithmic_MB=round(alg / 1e6, 2), achieved_GBs=round(alg / step_s / 1e9, 1),
                                 frac_of_hbm_peak=round(alg / step_s / 1e9 / HBM_PEAK_GBS, 4),
                                 what="(60 N + 40 M K) per scene + one read of img_feat, / ms_per_step / 8 TB/s")
    rep["site_us"] = {k: round(v, 2) for k, v in us.items()}
    return rep


def pipeline_bench(args, cfg, device):
    """BASELINE configs[3] ("EmbodiedScan mv-grounding config, full pipeline on 1xMI355X") as ONE chained call per step: B scenes of
    V = 50 synthetic 480 x 640 depth maps -> MultiViewIngest(100 000) -> forward(..., bbox=) with the shipped configuration's weights and
    fp32 features -> module.quantize(0.01) -> MinkResNet's four level coordinate sets -> batch_point_sample on the four 2D-backbone
    levels (DET:385-448, CFG:105-142).  Prints one JSON line with a `pipeline` object (ms per stage, ms total, scenes/s)."""
    import numpy as np
    from proxytransformation_amd.pipeline import MINK_RESNET_STRIDES, GroundingFeaturePrefix
    from proxytransformation_amd.synth import FPN_LEVELS, make_depth_scene
    B = args.scenes_per_gpu or 6
    mod, sd = build_module(cfg, device)
    V = cfg.V
    scenes_np = [make_depth_scene(cfg.seed_base + 50 + b, V=V) for b in range(B)]
    scenes = [dict(sc, depth_img=torch.from_numpy(sc["depth_img"].view(np.int16)).to(device).view(torch.uint16)) for sc in scenes_np]
    g = torch.Generator(device=device)
    g.manual_seed(cfg.seed_base)
    feats = [torch.randn((B, V, c, s_, s_), generator=g, device=device) for c, s_ in FPN_LEVELS]
    text = {"text_feats": torch.randn((B, cfg.L, cfg.embed_dim), generator=g, device=device),
            "text_token_mask": torch.ones((B, cfg.L), dtype=torch.bool, device=device)}
    pipe = GroundingFeaturePrefix(mod, n_points=cfg.N)
    with torch.no_grad():
        for i in range(3):
            res = pipe(scenes, text, feats, rng=np.random.RandomState(i))
        torch.cuda.synchronize()
        mod.check()
        # (a) the pixel draws of the two PointSample stages precomputed, as a dataloader worker would hand them over (the reference draws
        # them in its num_workers=6 loader processes, CFG:146): the chained call is device work + enqueue only
        pre = [dict(sc, choices=res.ingested.sel[b]) for b, sc in enumerate(scenes)]
        for i in range(args.warmup):
            pipe(pre, text, feats)
        torch.cuda.synchronize()
        blocks = []
        for r in range(max(1, args.repeats)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                res = pipe(pre, text, feats)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / args.steps)
        el = sorted(blocks)[len(blocks) // 2]
        # per-stage times: events on the caller's stream between the stages (they perturb the chain a little: separate calls)
        stage = {}
        for i in range(max(5, args.steps // 2)):
            r_ = pipe(pre, text, feats, time_stages=True)
            for k, v in r_.stage_ms.items():
                stage.setdefault(k, []).append(v)
        stage = {k: round(sorted(v)[len(v) // 2], 4) for k, v in stage.items()}
        # (b) the same call drawing the pixels itself on the host (np.random.choice per view + per scene, in the reference's order)
        t0 = time.perf_counter()
        nb = max(2, args.steps // 4)
        for i in range(nb):
            pipe(scenes, text, feats, rng=np.random.RandomState(i))
        torch.cuda.synchronize()
        el_draw = (time.perf_counter() - t0) / nb
        mod.check()
    depth_bytes = B * V * 480 * 640 * 2
    img_bytes = sum(int(f.numel()) * 4 for f in feats[-1:])
    nvox = int(res.coordinates.shape[0])
    lvl_rows = [sum(int(c.shape[0]) for c in lc) for lc in res.level_coords]
    alg = depth_bytes + img_bytes + B * (60 * cfg.N + 40 * cfg.M * cfg.num_sub) + B * cfg.N * 12 + nvox * 28 + \
        sum(n * (12 + 4 * FPN_LEVELS[li][0]) for li, n in enumerate(lvl_rows))
    line = dict(metric="scenes/sec (100k pts, 256 clusters, 64 proxies)", value=round(B / el, 2), unit="scenes/s", n_gpus=1,
                steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * el, 4), higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=f"{cfg.name} PIPELINE (BASELINE configs[3]): {B} scenes x {V} views of 480x640 uint16 depth -> ingest "
                                     f"{cfg.N} pts -> neck gs={cfg.grid_size} (M'={cfg.M_keep}, {cfg.text_blocks}+{cfg.img_blocks} blocks, "
                                     f"fp32 features) -> 1 cm voxels -> levels {MINK_RESNET_STRIDES} -> point sampling on "
                                     f"{[c for c, _ in FPN_LEVELS]}-channel maps", scenes_per_gpu=B, img_feat_dtype="f32"),
                pipeline=dict(ms_total=round(1e3 * el, 4), scenes_per_s=round(B / el, 2), ms_blocks=[round(1e3 * x, 4) for x in blocks],
                              ms_per_stage=stage,
                              with_host_pixel_draws=dict(ms_total=round(1e3 * el_draw, 3), scenes_per_s=round(B / el_draw, 2),
                                                         what="the call draws the PointSample indices itself (np.random.choice per view and "
                                                              "per scene on the host, the reference's order): dataloader-worker work in the "
                                                              "reference (CFG:146 num_workers=6)"),
                              voxel_rows=nvox, level_rows=lvl_rows, surviving_points=sum(int(o.shape[0]) for o in res.points),
                              what="ms_total: wall clock of `steps` chained calls between device synchronises / steps, pixel draws "
                                   "precomputed; ms_per_stage: HIP events between the stages of separate calls (median)"),
                roofline=dict(bound="hbm", kernel="whole chain", achieved=round(alg / el / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                              frac=round(alg / el / 1e9 / HBM_PEAK_GBS, 4), traffic=None, algorithmic_bytes_per_launch=alg,
                              what="depth maps (2 B/pixel) + one read of img_features[-1] + the neck's point side (60 N + 40 M K per "
                                   "scene) + voxel pass (12 N in, 28 B per row out) + level points and sampled features written; the "
                                   "chain is ~60 dependent launches: latency-bound, not a bandwidth figure"))
    if not args.no_cpu_baseline:
        try:
            from oracle import oracle
            from proxytransformation_amd.pipeline import projection_matrices
            threads = min(os.cpu_count() or 1, 32)
            sc = scenes_np[0]
            t0 = time.perf_counter()
            depth = sc["depth_img"].astype(np.float32) / np.float32(1000.0)
            ing = oracle.ingest(depth, sc["depth_cam2img"], sc["extrinsic"], cfg.N, rng=np.random.RandomState(0), num_threads=threads)
            f_last = feats[-1][:1].cpu().numpy()
            ref = oracle.forward(sd, grid_size=cfg.grid_size, dynamic_drop_radio=cfg.dynamic_drop_radio, num_sub=cfg.num_sub,
                                 num_heads=cfg.num_heads, text_blocks=cfg.text_blocks, img_blocks=cfg.img_blocks, points=ing["points"][None],
                                 text_feats=text["text_feats"][:1].cpu().numpy(), text_mask=text["text_token_mask"][:1].cpu().numpy(),
                                 img_feat=f_last, num_threads=threads)
            rc, _, _ = oracle.voxelize(ref["outputs"], 0.01)
            P = projection_matrices(sc["depth2img"])
            for li, s_ in enumerate(MINK_RESNET_STRIDES):
                lp = oracle.level_coordinates(rc, 1, s_)[0].astype(np.float32) * np.float32(0.01)
                oracle.point_sample(lp, feats[li][0].cpu().numpy(), P, scale=(0.75, 1.0), pad_hw=(480.0, 480.0))
            el_cpu = time.perf_counter() - t0
            line["cpu_baseline"] = dict(value=round(1.0 / el_cpu, 4), unit="scenes/s", cores=threads, kind="port",
                                        sample=f"ONE scene through the oracle chain (ingest -> forward -> voxelize -> levels -> point_sample; "
                                               f"includes the device-to-host copies of its feature maps) on {threads} of {os.cpu_count()} host threads")
        except Exception as e:
            line["cpu_baseline"] = dict(value=None, error=repr(e))
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))


def main():
    args = parse()
    if args.pipeline:
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
        if args.gpus != 1:
            raise SystemExit("--pipeline is a single-GPU line (scenes are independent: N ranks = N replicas)")
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        return pipeline_bench(args, CONFIGS[args.config], device)
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
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
    device = torch.device("cuda", 0 if args.share_gpu else local)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cfg = CONFIGS[args.config]
    B = args.scenes_per_gpu or cfg.B
    img_dtype = args.img_dtype or {"cfg2": "bf16", "cfg5": "f16"}.get(cfg.name, "f32")
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[img_dtype]
    mod, sd = build_module(cfg, device)
    inputs = InputSets(cfg, B, max(1, args.sets), rank, world, device, tdt)
    lib = _abi.lib()
    if args.gemm_policy is not None:
        lib.ptx_gemm_policy(args.gemm_policy)
    names = [lib.ptx_kernel_name(i).decode() for i in range(lib.ptx_kernel_count())]
    if args.time_kernel not in names:
        raise SystemExit(f"--time-kernel must be one of {names}")
    kid = names.index(args.time_kernel)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None
    workers = None
    extras = {}
    with torch.no_grad():
        for i in range(max(2 * len(inputs.sets), args.setup_forwards)):     # set-up, not warm-up: parameter tables, workspace,
            outs = mod(*inputs.args(i))             # pinned buffers; every input set is touched so that all of them are paged in
        torch.cuda.synchronize()
        for i in range(args.warmup):
            outs = mod(*inputs.args(i))
        if streams:
            for i in range(2 * len(streams)):               # every lane's context / workspace / probe, from this thread
                with torch.cuda.stream(streams[i % len(streams)]):
                    mod(*inputs.args(i))
            torch.cuda.synchronize()
            if not args.single_thread:
                workers = LaneWorkers(mod, inputs, streams)
                workers.run(0, 4 * len(streams))            # the threads' first calls
        n_out = sum(int(o.shape[0]) for o in outs)
        lib.ptx_timing_every(max(1, args.time_every))
        lib.ptx_timing_select(kid)
        # R blocks of exactly K steps; the input-set rotation continues across blocks
        block_s = []
        for r in range(max(1, args.repeats)):
            el, outs = timed_steps(mod, inputs, args.steps, barrier, streams, start=r * args.steps, workers=workers)
            block_s.append(el)
        if workers is not None:
            workers.close()
        launches, total_ms = ctypes.c_int(0), ctypes.c_float(0.0)
        lib.ptx_timing_read(ctypes.byref(launches), ctypes.byref(total_ms))
        lib.ptx_timing_select(-1)
        lib.ptx_timing_every(1)

        if not args.no_passes:
            steps2 = max(10, args.steps // 2)
            if tdt is not torch.float32:        # the same workload with fp32-stored image features (non-AMP layout)
                for i in range(4):
                    mod(*inputs.args(i, True))
                extras["f32"] = timed_steps(mod, inputs, steps2, barrier, None, True)[0] / steps2
            # opt-in reduced-precision compute (proxy-block GEMMs + attention on plain bf16 operands, fp32 accumulate): never the
            # headline; reported beside it with its error against the fp32 path on the same inputs
            ref_out = [o.clone() for o in mod(*inputs.args(0))]
            mod.compute_dtype = "bf16"
            for i in range(4):
                mod(*inputs.args(i))
            extras["bf16c"] = timed_steps(mod, inputs, steps2, barrier, None)[0] / steps2
            got = mod(*inputs.args(0))
            torch.cuda.synchronize()
            extras["bf16c_err"] = max(float((a - b).abs().max()) if a.shape == b.shape else float("inf")
                                      for a, b in zip(got, ref_out))
            mod.compute_dtype = "fp32"
            # the same step as a captured HIP graph (module.forward_padded: no host wait, padded outputs + device counts): ONE graph
            # holding one forward per input set, replayed back to back -- what a serving loop that does not need the lengths on the
            # host gets.  (r05: one graph PER input set replayed in rotation, the r04 form of this measurement, pays ~0.15 ms per
            # switch of graph executable on this stack and read 0.36-0.39 ms per forward: profiles/r05_graph_replay_forms.txt)
            try:
                gs = torch.cuda.Stream()
                nset = len(inputs.sets)
                with torch.cuda.stream(gs):
                    for j in range(nset):
                        mod.forward_padded(*inputs.args(j))
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=gs):
                    res = [mod.forward_padded(*inputs.args(j)) for j in range(nset)]
                for i in range(4):
                    g.replay()
                barrier()
                t0g = time.perf_counter()
                nrep = max(10, args.steps // nset)              # as many forwards as one timed block of the eager value
                for i in range(nrep):
                    g.replay()
                barrier()
                extras["graph"] = (time.perf_counter() - t0g) / (nrep * nset)
                del g, res
            except Exception as e:                              # never sinks the headline
                extras["graph_error"] = repr(e)
            # per-pass reports: single-rank runs only -- this block is rank 0's alone, so nothing in it may touch the process group
            # (its timed_steps get the local barrier); with N > 1 the other ranks would sit in the census all-gather meanwhile
            if rank == 0 and world == 1:
                pm = prefix_max(mod, inputs)
                extras["passes"] = [passes_report(cfg, B, site_times(lib, names, mod, inputs, 20), inputs.sets[0]["img"].element_size(),
                                                  pm, sorted(block_s)[len(block_s) // 2] / args.steps)]
                if cfg.name == "cfg2" and B * 8 <= 32:
                    wide = inputs.widened(8)
                    for i in range(3):
                        mod(*wide.args(i))
                    wide_us = site_times(lib, names, mod, wide, 9)
                    # the same two compute modes at 32 scenes per GPU (whole step, cold inputs)
                    t32 = {}
                    for cdt in ("fp32", "bf16"):
                        mod.compute_dtype = cdt
                        for i in range(3):
                            mod(*wide.args(i))
                        t32[cdt] = timed_steps(mod, wide, 9, torch.cuda.synchronize, None)[0] / 9
                    mod.compute_dtype = "fp32"
                    extras["wide"] = dict(scenes_per_gpu=wide.B, value_fp32_compute=round(wide.B / t32["fp32"], 2),
                                          value_bf16_compute=round(wide.B / t32["bf16"], 2))
                    extras["passes"].append(passes_report(cfg, wide.B, wide_us, wide.sets[0]["img"].element_size(), pm, t32["fp32"]))
                    del wide
        if not args.no_passes and rank == 0 and world == 1:
            with torch.enable_grad():
                try:
                    extras["train"] = train_step_ms(device)
                except Exception as e:                          # never sinks the eval number
                    extras["train"] = dict(ms=None, error=repr(e))
        if args.breakdown and rank == 0:
            print("per-kernel us/launch:", json.dumps({k: round(v, 2) for k, v in site_times(lib, names, mod, inputs, 12).items()}),
                  file=sys.stderr)

    ranks_seen = None
    if dist is not None:
        # every rank reports (rank, device it ran on): the line shows that the process group really had N members
        props = torch.cuda.get_device_properties(device)
        me = (rank, str(getattr(props, "uuid", "")) or f"{props.name}#{device.index}", os.getpid())
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        ranks_seen = sorted([list(x) for x in ranks_seen])
    vals = [extras.get("f32", 0.0), extras.get("bf16c", 0.0), extras.get("graph", 0.0)] + block_s
    t = torch.tensor(vals, device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # per block: the slowest rank
    f32_step, bf16c_step, graph_step, *block_s = (float(x) for x in t.tolist())
    elapsed = sorted(block_s)[len(block_s) // 2]           # the median block is the reported one

    if rank == 0:
        total_scenes = world * B * args.steps
        byts, _ = work_model(cfg, B, inputs.sets[0]["img"].element_size())
        abytes = byts.get(args.time_kernel)
        roof = None
        if launches.value > 0:
            avg_s = total_ms.value / launches.value / 1e3
            roof = dict(kernel=site_kernel(args.time_kernel, img_dtype), avg_launch_us=round(avg_s * 1e6, 2),
                        launches=launches.value)
            if abytes:
                ach = abytes / avg_s / 1e9
                kern = site_kernel(args.time_kernel, img_dtype)
                traffic, stale = pmc_traffic(kern, img_dtype, cfg, B)
                roof = dict(bound="hbm", kernel=kern, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_stale=stale,
                            traffic_source=os.path.relpath(PMC_FILE, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                           "this command; traffic_stale = taken with another build of the library)",
                            avg_launch_us=round(avg_s * 1e6, 2), launches=launches.value,
                            timed="HIP events attached to the kernel's own dispatch packet (hipExtLaunchKernel start / stop events: begin and end of "
                                  "the kernel itself, on its stream) on every %d-th launch of the measured steps" % max(1, args.time_every),
                            algorithmic_bytes_per_launch=abytes, so_sha16=so_sha16())
        line = dict(metric="scenes/sec (100k pts, 256 clusters, 64 proxies)", value=round(total_scenes / elapsed, 2),
                    unit="scenes/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                    ms_per_step=round(1e3 * elapsed / args.steps, 4), higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"{cfg.name}: N={cfg.N} pts, gs={cfg.grid_size}->M'={cfg.M_keep} kept clusters, "
                                         f"L={cfg.L} text + V={cfg.V} image proxies, d={cfg.embed_dim}, eval forward",
                                scenes_per_gpu=B, global_scenes_per_step=world * B, sharding="by scene, no collective",
                                img_feat_dtype=img_dtype, input_sets_rotated=len(inputs.sets), streams=args.streams,
                                lanes=("%d lanes: one host thread + torch stream each, forwards in flight side by side" % args.streams)
                                if args.streams > 1 else "1 (one forward at a time)",
                                setup_forwards=max(2 * len(inputs.sets), args.setup_forwards),
                                arithmetic="fp32 (fp32 MFMA / VALU; the 16-bit matrix pipe only through 3-way operand splits with fp32 accumulate: exact in the pooling pass, dropped terms <= 2^-25 |xy| in the 64x64-tile GEMMs)",
                                surviving_points_per_step=n_out),
                    roofline=roof)
        per_block = [round(world * B * args.steps / x, 2) for x in block_s]
        line["timed_blocks"] = dict(blocks=len(block_s), steps_per_block=args.steps, value_median=line["value"],
                                    value_min=min(per_block), value_max=max(per_block), values=per_block,
                                    what="each block = exactly `steps` forwards between barrier + device synchronise; "
                                         "`value` / `ms_per_step` are the median block's")
        if ranks_seen is not None:
            line["ranks_seen"] = ranks_seen
            line["backend"] = args.backend + (" (RCCL)" if args.backend == "nccl" else "")
        if f32_step > 0:
            line["value_f32_features"] = round(world * B / f32_step, 2)
            line["ms_per_step_f32_features"] = round(1e3 * f32_step, 4)
        if bf16c_step > 0:
            line["value_bf16_compute"] = round(world * B / bf16c_step, 2)
            line["bf16_compute"] = dict(ms_per_step=round(1e3 * bf16c_step, 4), max_abs_dxyz_vs_fp32=extras.get("bf16c_err"),
                                        what="compute_dtype='bf16': proxy-block GEMMs and attention on plain bf16 operands, fp32 "
                                             "accumulation; clustering / index tensors unchanged; not the headline value")
            if "wide" in extras:
                line["bf16_compute"]["at_32_scenes_per_gpu"] = extras["wide"]
        if graph_step > 0:
            line["value_graph_replay"] = round(world * B / graph_step, 2)
            line["graph_replay"] = dict(ms_per_step=round(1e3 * graph_step, 4),
                                        what="one HIP graph holding one forward per input set (module.forward_padded: padded outputs + device "
                                             "counts, no host wait) and replayed back to back; not the headline value")
        elif "graph_error" in extras:
            line["graph_replay"] = dict(error=extras["graph_error"])
        if "train" in extras:
            line["train_step_ms"] = extras["train"]["ms"]
            line["train_step"] = extras["train"]
        if "passes" in extras:
            line["roofline_passes"] = extras["passes"]
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
