"""CPU-only checks of the host side: C-ABI surface, registry / nn.Module boundary, layout rules,
scene sharding over a 2-rank gloo group.  No kernel is launched here."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "proxyt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from proxytransformation_amd import _abi
    lib = _abi.lib()
    syms = _header_symbols()
    assert len(syms) >= 15
    assert set(syms) == set(_abi.SIGNATURES), "binding table and include/proxyt.h disagree"
    for s in syms:
        getattr(lib, s)
    assert lib.ptx_abi_version() == _abi.ABI_VERSION
    # ... and NOTHING else: -fvisibility=hidden + csrc/exports.map keep the C++ internals and the kernel
    # handles out of the dynamic symbol table of both builds of the library
    import subprocess
    here = os.path.join(ROOT, "proxytransformation_amd")
    for name in ("libproxyt_hip.so", "libproxyt_hip_testhooks.so"):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(here, name)], check=True, capture_output=True,
                             text=True).stdout
        exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
        assert exported == sorted(syms), (name, sorted(set(exported) ^ set(syms))[:10])


def test_the_switchboard_of_the_product_library_is_four_variables():
    """VERDICT r04 #7 (no GPU needed): the product library reads exactly four environment variables -- each driven by a GPU test
    (tests/test_gpu_host.py) -- and carries no fault-injection hook; the hooks exist in the test-hooks build only, which in turn
    exports the same C ABI."""
    from proxytransformation_amd import _abi
    here = os.path.join(ROOT, "proxytransformation_amd")
    blob = open(os.path.join(here, "libproxyt_hip.so"), "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"PTX_[A-Z][A-Z_0-9]+", blob)))
    assert names == ["PTX_GATE", "PTX_GATE_TIMEOUT_MS", "PTX_LAYOUT", "PTX_POOL_NT"], names
    hooks = os.path.join(here, "libproxyt_hip_testhooks.so")
    assert os.path.exists(hooks), "make -C proxytransformation_amd/csrc builds both libraries"
    hblob = open(hooks, "rb").read()
    assert b"PTX_GATE_FAULT" in hblob and b"PTX_GATE_TRAP" in hblob
    h = ctypes.CDLL(hooks)
    for sname in _abi.SIGNATURES:
        getattr(h, sname)
    # the source agrees: getenv appears in the library only for these (and the two hooks under PTX_TEST_HOOKS)
    src = ""
    for fn in sorted(os.listdir(os.path.join(here, "csrc"))):
        if fn.endswith((".hip", ".h")):
            src += open(os.path.join(here, "csrc", fn)).read()
    read = sorted(set(re.findall(r'(?:getenv|env_on)\("(PTX_[A-Z_0-9]+)"\)', src)))
    assert read == ["PTX_GATE", "PTX_GATE_FAULT", "PTX_GATE_TIMEOUT_MS", "PTX_GATE_TRAP", "PTX_LAYOUT", "PTX_POOL_NT"], read


def test_struct_layout_matches_header():
    """ctypes mirrors must have the size the C compiler gives the header's structs."""
    from proxytransformation_amd import _abi
    prog = r'''
#include <stdio.h>
#include "proxyt.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(PtxShape), sizeof(PtxSlotMlp), sizeof(PtxBlock),
 sizeof(PtxBn1d), sizeof(PtxWeights), sizeof(PtxDebug), sizeof(PtxForwardOpts));return 0;}'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()
    got = [ctypes.sizeof(c) for c in (_abi.PtxShape, _abi.PtxSlotMlp, _abi.PtxBlock, _abi.PtxBn1d,
                                      _abi.PtxWeights, _abi.PtxDebug, _abi.PtxForwardOpts)]
    assert got == [int(x) for x in out]


def test_shape_validation_and_sizes_on_host():
    from proxytransformation_amd import MODELS, _abi
    lib = _abi.lib()
    m = MODELS.build(dict(type="ProxyTransformationNormReverse", grid_size=8, dynamic_drop_radio=0.5))
    s = m._shape(4, 100000, 64, 196)
    assert (s.Mt, s.Mk) == (359, 256)
    assert lib.ptx_workspace_bytes(ctypes.byref(s)) > 4 * 100000 * 4
    assert lib.ptx_prep_bytes(ctypes.byref(s)) > 0
    bad = m._shape(4, 100000, 64, 196)
    bad.C = 384                                   # 256 (the reference) and 512 (BASELINE configs[4]) only
    assert lib.ptx_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"embed_dim" in lib.ptx_last_error()
    big = MODELS.build(dict(type="ProxyTransformationNormReverse", grid_size=16, dynamic_drop_radio=0.75,
                            embed_dim=512))       # cfg5: the reference itself cannot run this
    s5 = big._shape(1, 500000, 64, 192, 2)
    assert (s5.Mt, s5.Mk, s5.C, s5.hidden) == (2868, 1024, 512, 2048)
    assert lib.ptx_workspace_bytes(ctypes.byref(s5)) > 0 and lib.ptx_prep_bytes(ctypes.byref(s5)) > 0
    assert big.textformer[0].attn.pc_bias.shape == (1, 1024, 23, 1)
    assert big.simple_encoder.mlp[0].weight.shape == (512, 6, 1, 1)
    bad = m._shape(4, 100000, 64, 196, 1)
    bad.hw = 289                                  # 16-bit features: at most 256 pixels -- rejected before any enqueue
    assert lib.ptx_workspace_bytes(ctypes.byref(bad)) == 0 and b"H*W" in lib.ptx_last_error()
    bad = m._shape(4, 100000, 64, 196)
    bad.K = 64
    assert lib.ptx_prep_bytes(ctypes.byref(bad)) == 0


def test_registry_and_constructor_surface():
    """Built exactly like the reference does (DET:95 with the dict of CFG:41)."""
    from proxytransformation_amd import MODELS, ProxyTransformationNormReverse
    cfg = dict(type="ProxyTransformationNormReverse", n_points=100000, grid_size=12, text_blocks=3,
               img_blocks=3, dynamic_drop_radio=0.6, num_sub=30)
    m = MODELS.build(cfg)
    assert isinstance(m, ProxyTransformationNormReverse) and isinstance(m, torch.nn.Module)
    assert m.num_cluster == 1728 and m.real_cluster_num == 691 and m.num_sub == 30
    import inspect
    sig = inspect.signature(ProxyTransformationNormReverse.__init__)
    want = dict(embed_dim=256, num_heads=8, n_points=100000, grid_size=4, text_blocks=1, img_blocks=1,
                dynamic_drop_radio=0.8, mlp_radio=4, qkv_bias=False, drop_rate=0.2, attn_drop_rate=0.2,
                drop_path_rate=0.2, num_sub=30, drop_radio=0.2, input_dim=512, img_spacial_dim=15)
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    fwd = inspect.signature(m.forward).parameters
    assert list(fwd)[:3] == ["points", "text_dict", "img_feat"]           # the reference's call (DET:385)
    assert all(p.default is not inspect.Parameter.empty for p in list(fwd.values())[3:])   # extras are optional


def test_state_dict_matches_reference_manifest():
    """Key names, shapes and dtypes equal the reference module's (manifest captured by
    tests/golden/gen_golden.py), so the authors' checkpoints load unchanged."""
    from proxytransformation_amd import MODELS
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))
    for entry in man:
        m = MODELS.build(dict(type="ProxyTransformationNormReverse", **entry["kwargs"]))
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _, _ in entry["tensors"]]
        for k, shape, dtype in entry["tensors"]:
            assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k


def test_no_cpu_fallback_and_error_classes():
    from tests.util import build_module
    from proxytransformation_amd.synth import PreshapeConfig, make_scene_batch
    cfg = PreshapeConfig("t", B=2, N=64, grid_size=4, dynamic_drop_radio=0.5, L=4, V=1)
    m, _ = build_module(cfg)
    pts, text, mask, img = make_scene_batch(cfg)
    td = {"text_feats": torch.from_numpy(text), "text_token_mask": torch.from_numpy(mask)}
    with pytest.raises(RuntimeError, match="no CPU path"):
        m([torch.from_numpy(p) for p in pts], td, torch.from_numpy(img))
    with pytest.raises(RuntimeError):                      # unequal N, like torch.cat at PRE:427
        m([torch.zeros(10, 3), torch.zeros(11, 3)], td, torch.from_numpy(img))
    m.train()                                              # train mode is HIP as well: same refusal of CPU tensors
    with pytest.raises(RuntimeError, match="no CPU path"):
        m([torch.from_numpy(p) for p in pts], td, torch.from_numpy(img))


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "proxytransformation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").lower() or f == "synth.py" and \
                    "import oracle" not in txt and "from oracle" not in txt, f"{f} mentions the oracle"
    code = ("import sys; sys.path.insert(0, %r); import proxytransformation_amd, bench; "
            "assert not any(k == 'oracle' or k.startswith('oracle.') for k in sys.modules)" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_synth_is_deterministic():
    from proxytransformation_amd.synth import CONFIGS, fill_tensor, make_scene_batch, PreshapeConfig
    a = fill_tensor("textformer.0.attn.qkv.weight", (768, 256))
    b = fill_tensor("textformer.0.attn.qkv.weight", (768, 256))
    assert np.array_equal(a, b) and a.dtype == np.float32 and abs(float(a.mean())) < 1e-3
    assert abs(float(a.var()) - 1 / 256) < 2e-4
    cfg = PreshapeConfig("t", B=3, N=100, grid_size=4, dynamic_drop_radio=0.5, L=6, V=1, seed_base=5)
    full = make_scene_batch(cfg)
    part = make_scene_batch(cfg, scene_ids=[2])
    assert np.array_equal(full[0][2], part[0][0]) and np.array_equal(full[3][2], part[3][0])
    assert full[2][1].tolist() == [True] * 4 + [False] * 2 and full[2][0].all()
    c2 = CONFIGS["cfg2"]
    assert (c2.M, c2.Mt, c2.M_keep, c2.Kd) == (512, 359, 256, 103)


# ------------------------------------------------------------------ sharding (world_size 2, gloo)
def test_scene_partition():
    from proxytransformation_amd.shard import scene_partition
    parts = scene_partition(32, 8)
    assert all(len(p) == 4 for p in parts) and sorted(sum(parts, [])) == list(range(32))
    assert scene_partition(5, 2) == [[0, 2, 4], [1, 3]]
    assert scene_partition(1, 4) == [[0], [], [], []]


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from proxytransformation_amd.shard import ShardedPreshape, gather_cluster_transforms, local_scene_ids
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
S, Mk = 5, 7
g = torch.Generator().manual_seed(0)
kc, tr, tf = torch.randn(S, Mk, 3, generator=g), torch.randn(S, Mk, 3, generator=g), torch.randn(S, Mk, 9, generator=g)
ids = local_scene_ids(S)
allp = gather_cluster_transforms(kc[ids], tr[ids], tf[ids], S)
ref = torch.cat([kc, tr, tf], -1)
assert torch.equal(allp, ref), "gathered transforms differ"
# a fake module that tags each scene so routing can be checked
class Fake:
    real_cluster_num = Mk
    def __call__(self, pts, td, img, return_transforms=False):
        f, m = td.values()
        outs = [p + f[i, 0, 0] + img[i, 0, 0, 0, 0] for i, p in enumerate(pts)]
        if not return_transforms:
            return outs
        sid = [int(p[0, 0]) for p in pts]
        return outs, dict(kcenter=kc[sid], translate=tr[sid], transform=tf[sid])
pts = [torch.full((4, 3), float(i)) for i in range(S)]
td = {"text_feats": torch.arange(S).float().view(S, 1, 1) * 10, "text_token_mask": torch.ones(S, 1, dtype=torch.bool)}
img = torch.arange(S).float().view(S, 1, 1, 1, 1) * 100
sp = ShardedPreshape(Fake())
lids, outs = sp(pts, td, img)
assert lids == ids
for i, o in zip(lids, outs):
    assert torch.equal(o, torch.full((4, 3), float(i + 10 * i + 100 * i)))
# rank-local inputs (what a per-rank dataloader hands over) + the transform gather
sel = torch.tensor(ids)
ltd = {"text_feats": td["text_feats"][sel], "text_token_mask": td["text_token_mask"][sel]}
lids2, outs2, allt = sp([pts[i] for i in ids], ltd, img[sel], inputs="local", num_scenes=S, gather=True)
assert lids2 == ids and all(torch.equal(a, b) for a, b in zip(outs, outs2))
assert torch.equal(allt, ref), "gathered transforms (local inputs) differ"
try:
    sp(pts, td, img, inputs="local", num_scenes=S)
    raise SystemExit("local inputs of the wrong length were accepted")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharding_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
        assert "ok" in out


def test_dropout_site_seeds_are_distinct():
    """Every dropout / DropPath site of the two live ProxyBlocks, and the second mask of each attention node, has its
    own seed (train.site_seeds); calls and module instances are separated too."""
    from proxytransformation_amd import train
    seen = set()
    for call in (1, 2, 3):
        for salt in (1, 2):
            seeds = train.site_seeds(42, call, salt)
            flat = [s for br in seeds for s in br] + [br[0] + 1 for br in seeds]
            assert len(set(flat)) == 14
            assert not (seen & set(flat))
            seen |= set(flat)


def test_module_copies_drop_the_host_caches():
    """copy.deepcopy / torch.save of a module that has run (ctypes pointers in its caches) must work and must not share
    library contexts (r02 advisory)."""
    import copy
    import ctypes
    import io
    from proxytransformation_amd import MODELS
    m = MODELS.build(dict(type="ProxyTransformationNormReverse", grid_size=4))
    m._wstruct = ctypes.c_void_p(5)             # what a forward leaves behind
    m._shapes[(1, 2)] = ctypes.c_void_p(7)
    m2 = copy.deepcopy(m)
    assert m2._wstruct is None and m2._lanes == {} and m2._shapes == {} and m2._instance_salt != m._instance_salt
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert list(m3.state_dict()) == list(m.state_dict()) and m3._wstruct is None


def test_pipeline_host_pieces():
    """The host-side pieces of the configs[3] chain (proxytransformation_amd/pipeline.py), no GPU: the synthetic depth scan
    back-projects into its room through the ORACLE's ingest, projection matrices are intrinsic @ extrinsic, the oracle's level
    coordinates are the distinct floor(c / s) * s in first-occurrence order (negative coordinates floor, not truncate)."""
    from oracle import oracle
    from proxytransformation_amd.pipeline import MINK_RESNET_STRIDES, projection_matrices
    from proxytransformation_amd.synth import FPN_LEVELS, make_depth_scene
    sc = make_depth_scene(11, V=3, H=120, W=160)
    assert sc["depth_img"].dtype == np.uint16 and sc["depth_img"].shape == (3, 120, 160)
    depth = sc["depth_img"].astype(np.float32) / np.float32(sc["depth_shift"])
    r = oracle.ingest(depth, sc["depth_cam2img"], sc["extrinsic"], 5000, rng=np.random.RandomState(0))
    assert (r["points"] > -5e-3).all() and (r["points"] < np.array([7, 5, 3], np.float32) + 5e-3).all()
    P = projection_matrices(sc["depth2img"])
    assert P.shape == (3, 4, 4) and np.allclose(P[1], sc["depth2img"]["intrinsic"][1] @ sc["depth2img"]["extrinsic"][1])
    # a point of view 0 projects back into view 0's image with positive depth
    q = P[0] @ np.append(r["points"][0], 1.0)
    assert MINK_RESNET_STRIDES == (8, 16, 32, 64) and [c for c, _ in FPN_LEVELS] == [64, 128, 256, 512]
    coords = np.array([[0, 9, -1, 17], [0, 15, -8, 23], [0, 16, -9, 0], [1, 9, -1, 17], [1, 1, 1, 1]], np.int32)
    lv = oracle.level_coordinates(coords, 2, 8)
    assert np.array_equal(lv[0], [[8, -8, 16], [16, -16, 0]]) and np.array_equal(lv[1], [[8, -8, 16], [0, 0, 0]])


def test_ctypes_structs_match_the_header(tmp_path):
    """Every struct that crosses the C ABI by pointer: size and the offset of every field as gcc lays out include/proxyt.h must
    equal what ctypes lays out for proxytransformation_amd/_abi.py (a field added on one side only would shift everything behind
    it silently)."""
    from proxytransformation_amd import _abi
    structs = {"PtxShape": _abi.PtxShape, "PtxTrainBlock": _abi.PtxTrainBlock, "PtxTrainImgPool": _abi.PtxTrainImgPool,
               "PtxTrainSlotNet": _abi.PtxTrainSlotNet, "PtxTrainStepLayout": _abi.PtxTrainStepLayout, "PtxTrainStep": _abi.PtxTrainStep,
               "PtxForwardOpts": _abi.PtxForwardOpts, "PtxWeights": _abi.PtxWeights, "PtxBlock": _abi.PtxBlock}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "proxyt.h")}"', "int main(void) {"]
    for name, cls in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ['printf("NGRAD %d\\n", (int)PTX_TS_NGRAD);', "return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(out[name]) == ctypes.sizeof(cls), (name, out[name], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"
    assert int(out["NGRAD"]) == _abi.TS_NGRAD
