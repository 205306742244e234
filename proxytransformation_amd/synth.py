"""Synthetic scenes and closed-form deterministic weights for the preshape path.

The reference publishes no checkpoints, fixtures or datasets that can travel
(SURVEY.md section 4), so every test / bench input is generated from seeds:

* scenes follow SURVEY.md section 8d: ``points = U[0,1)^3 * extent`` in generation
  order (the reference shuffles points, transforms/points.py:411), text proxies
  ``N(0,1)``, token mask all-true except scene 1 whose last ``L//3`` tokens are
  padding, image features ``N(0,1)`` of shape (B, V, input_dim, H, W);
* weights are a counter-based hash of (parameter name, flat index) so that a
  2.18 M-parameter ``state_dict`` never has to be stored: the golden-vector
  generator, the oracle and the HIP module all rebuild the same tensors bit for
  bit on any machine (pure uint64/float64 numpy arithmetic, no RNG state).

numpy's PCG64 ``Generator`` is used for the inputs because its stream is stable
across numpy releases; torch's CPU generator is not guaranteed to be.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np

__all__ = [
    "PreshapeConfig", "CONFIGS", "make_scene_batch", "fill_tensor", "fill_state_dict", "make_depth_scene",
    "FPN_LEVELS",
]


@dataclass(frozen=True)
class PreshapeConfig:
    """One row of SURVEY.md section 8's config table (constructor kwargs + input shape)."""
    name: str
    B: int
    N: int
    grid_size: int
    dynamic_drop_radio: float      # sic: the reference spells it "radio" (PRE:282)
    L: int                         # text proxies
    V: int                         # image proxies (views)
    embed_dim: int = 256
    num_heads: int = 8
    num_sub: int = 30
    input_dim: int = 512
    img_spacial_dim: int = 15      # sic (PRE:285)
    text_blocks: int = 1
    img_blocks: int = 1
    extent: Tuple[float, float, float] = (12.0, 12.0, 9.0)
    seed_base: int = 0
    #: "uniform": points = U[0,1)^3 * extent; "two_blob": every point in one of two balls of radius 3 m in opposite
    #: corners of the extent box -- most grid centres never fill their slots, so both ball queries scan the WHOLE scene (the regime
    #: in which the clustering pass is a bandwidth kernel, VERDICT r05 "next" #5)
    distribution: str = "uniform"

    @property
    def M(self) -> int:
        return self.grid_size ** 3

    @property
    def Mt(self) -> int:            # after the 0.3 empty-drop (PRE:374-376)
        return self.M - int(self.M * 0.3)

    @property
    def M_keep(self) -> int:        # PRE:389 / PRE:195
        return int(self.M * (1 - self.dynamic_drop_radio))

    @property
    def Kd(self) -> int:            # FPS picks = clusters to drop (PRE:390)
        return self.Mt - self.M_keep

    def module_kwargs(self) -> dict:
        return dict(embed_dim=self.embed_dim, num_heads=self.num_heads, n_points=self.N,
                    grid_size=self.grid_size, text_blocks=self.text_blocks,
                    img_blocks=self.img_blocks, dynamic_drop_radio=self.dynamic_drop_radio,
                    num_sub=self.num_sub, input_dim=self.input_dim,
                    img_spacial_dim=self.img_spacial_dim)


# BASELINE.json "configs" mapped to constructor arguments.
CONFIGS: Dict[str, PreshapeConfig] = {
    # cfg1: reference's own CPU-runnable plumbing case
    "cfg1": PreshapeConfig("cfg1", B=1, N=20000, grid_size=8, dynamic_drop_radio=0.875,
                           L=16, V=4, seed_base=1000),
    # cfg2: the configuration the metric is quoted on (100k pts, 256 kept clusters,
    # 64 text + 196 image proxies, d=256); B is the per-GPU shard of cfg3 (4 scenes)
    "cfg2": PreshapeConfig("cfg2", B=4, N=100000, grid_size=8, dynamic_drop_radio=0.5,
                           L=64, V=196, seed_base=2000),
    # cfg4: the reference's only shipped config (CFG:41): gs=12, ddr=0.6, 3+3 blocks
    "cfg4": PreshapeConfig("cfg4", B=1, N=100000, grid_size=12, dynamic_drop_radio=0.6,
                           L=20, V=50, text_blocks=3, img_blocks=3, seed_base=4000),
    # cfg4 in the regime a real ScanNet room puts it in (SURVEY 8d's second distribution): every side of a (7, 5, 3) m room is
    # shorter than 2 * margin = 8 m, so PRE:48 yields an inverted grid, PRE:62 clamps it and both ball queries fill their 30 slots
    # from the first few hundred points of the scene (no padded slot anywhere, ~100 distinct clustered points per scene)
    "cfg4_room": PreshapeConfig("cfg4_room", B=1, N=100000, grid_size=12, dynamic_drop_radio=0.6,
                                L=20, V=50, text_blocks=3, img_blocks=3, extent=(7.0, 5.0, 3.0), seed_base=4000),
    # cfg5: stress / roofline run; the reference itself cannot run d=512
    "cfg5": PreshapeConfig("cfg5", B=1, N=500000, grid_size=16, dynamic_drop_radio=0.75,
                           L=64, V=192, embed_dim=512, seed_base=5000),
}


def make_scene_batch(cfg: PreshapeConfig, scene_ids=None, *, mask_scene: int = 1):
    """Return numpy inputs for the scenes ``scene_ids`` (default ``range(cfg.B)``).

    Scene ``i`` depends only on ``cfg.seed_base + i`` so that any rank of a
    sharded run can build exactly its own scenes.
    Returns ``points (b,N,3) f32, text_feats (b,L,C) f32, text_mask (b,L) bool,
    img_feat (b,V,input_dim,H,W) f32``.
    """
    if scene_ids is None:
        scene_ids = range(cfg.B)
    scene_ids = list(scene_ids)
    b = len(scene_ids)
    hw = cfg.img_spacial_dim
    points = np.empty((b, cfg.N, 3), np.float32)
    text = np.empty((b, cfg.L, cfg.embed_dim), np.float32)
    mask = np.ones((b, cfg.L), np.bool_)
    img = np.empty((b, cfg.V, cfg.input_dim, hw, hw), np.float32)
    ext = np.asarray(cfg.extent, np.float32)
    for j, sid in enumerate(scene_ids):
        rng = np.random.default_rng(cfg.seed_base + int(sid))
        if cfg.distribution == "uniform":
            points[j] = rng.random((cfg.N, 3), dtype=np.float32) * ext
        elif cfg.distribution == "two_blob":
            d = rng.standard_normal((cfg.N, 3)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True) + np.float32(1e-12)
            r = np.float32(3.0) * np.cbrt(rng.random((cfg.N, 1), dtype=np.float32))
            corner = np.where(rng.random((cfg.N, 1)) < 0.5, np.float32(3.0), ext[None, :] - np.float32(3.0)).astype(np.float32)
            points[j] = corner + d * r
        else:
            raise ValueError(f"distribution {cfg.distribution!r}")
        text[j] = rng.standard_normal((cfg.L, cfg.embed_dim), dtype=np.float32)
        img[j] = rng.standard_normal((cfg.V, cfg.input_dim, hw, hw), dtype=np.float32)
        if int(sid) == mask_scene and cfg.L >= 3:
            mask[j, cfg.L - cfg.L // 3:] = False
    return points, text, mask, img


# --------------------------------------------------------------------------
# synthetic multi-view depth scenes (BASELINE configs[3]: the shipped pipeline, CFG:105-142)
# --------------------------------------------------------------------------
#: (channels, side) of the 2D backbone's four output levels for a 480 x 480 input: mmdet ResNet-50 with base_channels=16
#: (CFG:28-39) has stage widths 64 / 128 / 256 / 512 at strides 4 / 8 / 16 / 32; the last one is the neck's img_feat (DET:385)
FPN_LEVELS = ((64, 120), (128, 60), (256, 30), (512, 15))


def make_depth_scene(seed: int, V: int = 50, H: int = 480, W: int = 640, extent=(7.0, 5.0, 3.0), *, as_u16: bool = True,
                     hole_rate: float = 0.08):
    """One synthetic RGB-D scan of a box-shaped room of size ``extent`` (metres, one corner at the origin): ``V`` pinhole cameras at
    random poses inside the room look at its walls / floor / ceiling through "room-sized frusta" -- every pixel's depth is the
    distance (along the optical axis) to the first surface its ray meets, shortened inside smooth random blobs (furniture) and
    zero (invalid, like a sensor hole) for ``hole_rate`` of the pixels.  Back-projected through ``depth_cam2img`` and the
    global2cam ``extrinsic`` the pixels land inside ``[0, extent]`` -- the (7, 5, 3) m regime of SURVEY 8d.

    Returns a dict shaped like what the reference's loading transforms hand on (mv_3dvg_dataset.py:535-553,
    transforms/multiview.py:140-190): ``depth_img`` (V,H,W) uint16 raw (``depth_shift`` 1000) or float32 metres,
    ``depth_cam2img`` (4,4), ``depth2img = dict(extrinsic=[V x (4,4) global2cam], intrinsic=[V x (4,4) cam2img])`` and an
    ``img_meta`` for a 480 x 640 image resized to 480 x 480 (CFG:114: ``scale_factor`` (0.75, 1.0))."""
    rng = np.random.default_rng(seed)
    ex = np.asarray(extent, np.float64)
    fx = fy = 577.0 * W / 640.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, cx, cy
    us, vs = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    dc = np.stack([(us - cx) / fx, (vs - cy) / fy, np.ones_like(us)], -1).astype(np.float32)         # (H,W,3), z = 1
    depth = np.empty((V, H, W), np.float32)
    extr = []
    for v in range(V):
        pos = ex * (0.15 + 0.7 * rng.random(3))
        yaw, pitch = 2 * np.pi * rng.random(), 0.5 * (rng.random() - 0.5)
        fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])
        right = np.cross(fwd, [0.0, 0.0, 1.0])
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], 1)                                   # cam2global rotation (columns = camera axes)
        dw = dc @ R.T.astype(np.float32)                                      # ray directions in the room's frame
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.full((H, W), np.inf, np.float32)
            for a in range(3):
                da = dw[..., a]
                ta = np.where(da > 0, (ex[a] - pos[a]) / da, np.where(da < 0, (0.0 - pos[a]) / da, np.inf))
                t = np.minimum(t, ta.astype(np.float32))
        # furniture: a few smooth blobs in the image plane pull the surface towards the camera
        for _ in range(4):
            bu, bv, br = W * rng.random(), H * rng.random(), (0.08 + 0.15 * rng.random()) * W
            w = np.exp(-(((us - bu) ** 2 + (vs - bv) ** 2) / (2 * br * br))).astype(np.float32)
            t = t * (1.0 - np.float32(0.45 * rng.random()) * w)
        t[rng.random((H, W)) < hole_rate] = 0.0
        depth[v] = t
        c2g = np.eye(4)
        c2g[:3, :3], c2g[:3, 3] = R, pos
        extr.append(np.linalg.inv(c2g).astype(np.float32))
    raw = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
    img_meta = dict(scale_factor=(480.0 / W, 480.0 / H), img_shape=(480, 480), flip=False)
    return dict(depth_img=raw if as_u16 else raw.astype(np.float32) / np.float32(1000.0), depth_shift=1000.0, depth_cam2img=K,
                depth2img=dict(extrinsic=extr, intrinsic=[K.copy() for _ in range(V)]), extrinsic=np.stack(extr),
                img_meta=img_meta)


# --------------------------------------------------------------------------
# deterministic weights
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised SplitMix64 finaliser on uint64 (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform_pm1(name: str, numel: int, salt: int = 0) -> np.ndarray:
    """``numel`` float64 values in [-1, 1) that depend only on (name, index, salt)."""
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(numel, dtype=np.uint64)
               + (key << np.uint64(32)) + np.uint64(salt & 0xFFFF))
    bits = _splitmix64(ctr) >> np.uint64(11)                 # 53 random bits
    return bits.astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fill_tensor(name: str, shape, salt: int = 0) -> np.ndarray:
    """Closed-form value for the state_dict entry ``name`` of shape ``shape``.

    Scales are chosen so every stage of the path is exercised in a realistic
    numeric range (non-saturated tanh offsets, non-trivial BN running stats,
    O(1) normalised transforms); they are not the reference's init.
    """
    shape = tuple(int(s) for s in shape)
    numel = int(np.prod(shape)) if shape else 1
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    u = _uniform_pm1(name, numel, salt)
    if leaf == "running_var":
        v = 1.0 + 0.5 * u                                    # [0.5, 1.5)
    elif leaf == "running_mean":
        v = 0.1 * u
    elif leaf == "positional_embedding":
        v = u * (3.0 / shape[-1]) ** 0.5
    elif leaf in ("pb_bias", "pc_bias", "pr_bias"):
        v = 0.05 * u
    elif len(shape) == 1:
        # norm weights ~1, every bias small
        v = (1.0 + 0.1 * u) if leaf == "weight" else 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        v = u * (3.0 / fan_in) ** 0.5                        # var = 1/fan_in
    return v.reshape(shape).astype(np.float32)


def fill_state_dict(template: Dict[str, "object"], salt: int = 0) -> Dict[str, np.ndarray]:
    """Map ``{name: tensor-like with .shape}`` to deterministic numpy arrays."""
    return {k: fill_tensor(k, tuple(v.shape), salt) for k, v in template.items()}
