#!/bin/bash
# which library kernels torch.matmul picks for the cfg-2 fp32 GEMM shapes (names carry the macro tile): rocprofv3 --kernel-trace
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/tgn
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tgn -o t -- python $R/tools/gemm_vs_torch.py > /tmp/tgn.log 2>&1
python - <<'PYEOF'
import csv, glob, collections
f = glob.glob("/tmp/tgn/*kernel_trace.csv")[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.startswith("Cijk") or "gemm" in n.lower() and "ytvln" not in n:
        k = (n[:230], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"))
        d = acc.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, g, w, lds, vg), (c, t) in acc.items():
    print(f"calls {c:3d} avg {t / c:8.1f} us grid {g} wg {w} lds {lds} vgpr {vg}  {n}")
PYEOF
