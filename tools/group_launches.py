"""The grouped weight-gradient launches of ONE replayed step of bench.py's kernel trace, in launch order: kernel, workgroups, us.
    python tools/group_launches.py <bench_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "stem_pack" in r["Kernel_Name"]]
step = rows[first[-2]:first[-1]]
for r in step:
    n = r["Kernel_Name"]
    if "gemm_group_kernel" in n or ("gemm_kernel" in n and "KrowTr" in n.split("KrowTr")[0] + "KrowTr" and n.count("KrowTr") >= 2):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
        kind = "group" if "gemm_group_kernel" in n else "single"
        gather = "gather" if "ELi2EEE" in n.split("KrowTr")[-1][:12] else "plain"
        tile = "128x128" if "Li128ELi128E" in n else ("64x64" if "Li64ELi64E" in n else "?")
        print(f"{kind:6s} {tile:8s} {gather:6s} workgroups {wg:6d}  {us:8.1f} us   lds {r.get('LDS_Block_Size', '?')} vgpr {r.get('VGPR_Count', '?')}")
