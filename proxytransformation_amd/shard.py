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
    pad = packed.new_zeros((bmax, Mk, 15))
    pad[: packed.shape[0]] = packed
    gathered = packed.new_empty((world, bmax, Mk, 15))
    dist.all_gather_into_tensor(gathered.view(world * bmax, Mk, 15), pad, group=group)
    out = packed.new_empty((num_scenes, Mk, 15))
    for r, ids in enumerate(parts):
        for j, sid in enumerate(ids):
            out[sid] = gathered[r, j]
    return out


class ShardedPreshape:
    """Run a (replicated) preshape module on this rank's scenes of a global batch."""

    def __init__(self, module, rank: Optional[int] = None, world_size: Optional[int] = None):
        self.module = module
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)

    def local_ids(self, num_scenes: int) -> List[int]:
        return scene_partition(num_scenes, self.world_size)[self.rank]

    def __call__(self, points: Sequence[torch.Tensor], text_dict: dict, img_feat: torch.Tensor):
        """Inputs are the GLOBAL batch (or any indexable holding at least the local scenes);
        returns ``(local_ids, outputs_for_local_scenes)``."""
        ids = self.local_ids(len(points))
        if not ids:
            return ids, []
        sel = torch.as_tensor(ids, device=img_feat.device)
        feats, mask = text_dict.values()
        local_text = {"text_feats": feats.index_select(0, sel), "text_token_mask": mask.index_select(0, sel)}
        outs = self.module([points[i] for i in ids], local_text, img_feat.index_select(0, sel))
        return ids, outs
