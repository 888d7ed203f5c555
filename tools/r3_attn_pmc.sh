#!/bin/bash
# counters of the attention kernels of the final round-3 tree (image self-attention; co-attention)
bash tools/attn_pmc.sh round3_img img > /dev/null 2>&1
bash tools/attn_pmc.sh round3_co co > /dev/null 2>&1
python - <<'PY'
import json
for t in ("round3_img", "round3_co"):
    d = json.load(open("gpurun_out/%s_attn_pmc.json" % t))
    for k, v in d.items():
        print(t, k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if not kk.startswith("SQ_") and not kk.startswith("GRBM")})
PY
