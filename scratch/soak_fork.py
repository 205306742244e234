"""Fork / join soak for the device-word gates: the inputs of every forward are (re)written IN PLACE on the caller's stream right
before the call (copies from three resident variants into the same tensors), so a clustering stream that started too early, or
read stale lines, gives different indices / coordinates than the first pass over that variant; outputs compared bitwise.
python scratch/soak_fork.py [config] [forwards] [f32]   (SOAK_B=<scenes per call>)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_module, InputSets
from proxytransformation_amd.synth import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
cfg = CONFIGS[name]; dev = torch.device("cuda:0")
if os.environ.get("SOAK_B"):                                  # scenes per call (default: the configuration's own)
    import dataclasses
    cfg = dataclasses.replace(cfg, B=int(os.environ["SOAK_B"]))
mod, _ = build_module(cfg, dev)
inp = InputSets(cfg, cfg.B, 3, 0, 1, dev, torch.float32 if f32 else torch.bfloat16)
variants = [inp.args(k) for k in range(3)]
pts = [p.clone() for p in variants[0][0]]
text = {k: v.clone() for k, v in variants[0][1].items()}
img = variants[0][2].clone()
ref, bad = {}, 0
with torch.no_grad():
    for i in range(n):
        v = variants[(i * 7 + i // 5) % 3]
        for d, s_ in zip(pts, v[0]): d.copy_(s_)             # caller-stream work the forward has to be ordered behind
        text["text_feats"].copy_(v[1]["text_feats"])
        img.copy_(v[2])
        outs = mod(pts, text, img)
        key = (i * 7 + i // 5) % 3
        if key not in ref:
            ref[key] = [o.clone() for o in outs]
        else:
            for a, b in zip(outs, ref[key]):
                if a.shape != b.shape or not torch.equal(a, b):
                    bad += 1
                    break
torch.cuda.synchronize()
print(name, "f32" if f32 else "bf16", "forwards", n, "mismatching", bad, "PTX_GATE", os.environ.get("PTX_GATE"))
