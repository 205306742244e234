import json, glob, os, sys
O = sys.argv[1]
for f in sorted(glob.glob(os.path.join(O, "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "ERR", e); continue
    extra = ""
    if "bf16_compute" in d and "at_32_scenes_per_gpu" in d["bf16_compute"]:
        extra += f"  @32: {d['bf16_compute']['at_32_scenes_per_gpu']['value_fp32_compute']}"
    if d.get("train_step_ms"):
        extra += f"  train {d['train_step_ms']} ms {d['train_step'].get('ms_blocks')}"
    if "roofline_passes" in d and len(d["roofline_passes"]) > 1:
        r = d["roofline_passes"][1]
        extra += f"  gemms@32 {r['block_gemms_mfma']['achieved_TFLOPs']} TF ({r['block_gemms_mfma']['us']} us)"
    print(f"{os.path.basename(f):28s} value {d['value']:10.1f}  ms {d['ms_per_step']}{extra}")
