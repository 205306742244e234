"""Scene sharding of the preshape path across the GPUs of one node.

In eval mode every scene is independent (BatchNorm uses running statistics and
nothing in PRE:424-469 mixes scenes), so the path shards by scene with NO
data-path collective: rank ``r`` of ``W`` owns scenes ``r, r+W, r+2W, ...``
(round-robin keeps the per-rank count within one of each other).  This is the
reference's own deployment shape: plain DDP, one process per GPU, each rank
preshaping its own mini-batch (configs/default_runtime.py:15, README.md:83).

The only exchange the path may need is optional and tiny: callers that want the
per-cluster transforms of *all* scenes on every rank (e.g. for logging or for a
downstream stage sharded differently) gather ``(centre | translate | transform)``
= 15 floats per kept cluster with ONE all-gather -- with backend ``nccl`` that is
RCCL over xGMI; at <= tens of KB per rank it is latency-bound, so a single
one-shot all-gather is used rather than bucketed or ring-chunked traffic.
Transformed point clouds are never exchanged (variable length, 1.2 MB/scene).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

__all__ = ["scene_partition", "local_scene_ids", "gather_cluster_transforms", "ShardedPreshape"]


def scene_partition(num_scenes: int, world_size: int) -> List[List[int]]:
    """Round-robin assignment scene -> rank."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    return [list(range(r, num_scenes, world_size)) for r in range(world_size)]


def local_scene_ids(num_scenes: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> List[int]:
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return scene_partition(num_scenes, world_size)[rank]


def gather_cluster_transforms(kcenter: torch.Tensor, translate: torch.Tensor, transform: torch.Tensor,
                              num_scenes: int, group=None) -> torch.Tensor:
    """All-gather the per-cluster affine parameters of the local scenes.

    kcenter (b,Mk,3), translate (b,Mk,3), transform (b,Mk,9) of this rank's scenes (in
    ``local_scene_ids`` order) -> (num_scenes, Mk, 15) in global scene order on every rank.
    Ranks may own different numbers of scenes; shards are padded to the maximum for the
    collective and un-padded afterwards.
    """
    packed = torch.cat([kcenter, translate, transform.reshape(transform.shape[0], transform.shape[1], 9)], dim=-1)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return packed
    world = dist.get_world_size(group)
    parts = scene_partition(num_scenes, world)
    bmax = max(len(p) for p in parts)
    Mk = packed.shape[1]
    # gloo (CPU tests, two ranks sharing one GPU) moves host tensors; nccl (= RCCL over xGMI) device tensors
    via_host = packed.is_cuda and dist.get_backend(group) == "gloo"
    src = packed.cpu() if via_host else packed
    pad = src.new_zeros((bmax, Mk, 15))
    pad[: src.shape[0]] = src
    gathered = src.new_empty((world, bmax, Mk, 15))
    dist.all_gather_into_tensor(gathered.view(world * bmax, Mk, 15), pad, group=group)
    if via_host:
        gathered = gathered.to(packed.device)
    out = packed.new_empty((num_scenes, Mk, 15))
    for r, ids in enumerate(parts):
        if ids:
            out[torch.as_tensor(ids, device=out.device)] = gathered[r, : len(ids)]
    return out


class ShardedPreshape:
    """Run a (replicated) preshape module on this rank's scenes of a global batch.

    Two input conventions:

    * ``inputs="local"`` (the deployment shape: every rank's dataloader produced only its own scenes --
      plain DDP, README.md:83): ``points`` / ``text_dict`` / ``img_feat`` hold exactly this rank's scenes in
      ``local_ids(num_scenes)`` order; nothing global is ever resident on a rank (cfg3: 4 of 32 scenes,
      0.36 GB instead of 2.9 GB of image features);
    * ``inputs="global"``: the tensors hold the whole batch (convenient for tests on one device) and this
      rank's scenes are selected from them.

    ``gather=True`` additionally all-gathers the per-cluster transforms of ALL scenes
    (``gather_cluster_transforms``: one collective of 15 floats per kept cluster)."""

    def __init__(self, module, rank: Optional[int] = None, world_size: Optional[int] = None, group=None):
        self.module = module
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)

    def local_ids(self, num_scenes: int) -> List[int]:
        return scene_partition(num_scenes, self.world_size)[self.rank]

    def __call__(self, points: Sequence[torch.Tensor], text_dict: dict, img_feat: torch.Tensor, *,
                 inputs: str = "global", num_scenes: Optional[int] = None, gather: bool = False):
        """Returns ``(local_ids, outputs_for_local_scenes)``, plus the gathered ``(num_scenes, M', 15)``
        transforms when ``gather=True``."""
        if inputs not in ("global", "local"):
            raise ValueError('inputs must be "global" or "local"')
        if inputs == "global":
            num_scenes = len(points) if num_scenes is None else num_scenes
        elif num_scenes is None:
            raise ValueError('inputs="local" needs num_scenes (the size of the global batch)')
        ids = self.local_ids(num_scenes)
        feats, mask = text_dict.values()
        if inputs == "local":
            if len(points) != len(ids):
                raise ValueError(f"rank {self.rank} owns {len(ids)} of {num_scenes} scenes but got {len(points)}")
            local_pts, local_text, local_img = list(points), {"text_feats": feats, "text_token_mask": mask}, img_feat
        elif ids:
            sel = torch.as_tensor(ids, device=img_feat.device)
            local_pts = [points[i] for i in ids]
            local_text = {"text_feats": feats.index_select(0, sel), "text_token_mask": mask.index_select(0, sel)}
            local_img = img_feat.index_select(0, sel)
        outs, tf = [], None
        if ids:
            if gather:
                outs, tf = self.module(local_pts, local_text, local_img, return_transforms=True)
            else:
                outs = self.module(local_pts, local_text, local_img)
        if not gather:
            return ids, outs
        Mk = self.module.real_cluster_num
        dev = img_feat.device
        if tf is None:          # a rank without scenes still takes part in the collective
            tf = dict(kcenter=torch.zeros((0, Mk, 3), device=dev), translate=torch.zeros((0, Mk, 3), device=dev),
                      transform=torch.zeros((0, Mk, 9), device=dev))
        allt = gather_cluster_transforms(tf["kcenter"], tf["translate"], tf["transform"], num_scenes, self.group)
        return ids, outs, allt
