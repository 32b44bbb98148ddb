import csv, json, sys, collections
calls = json.load(open(sys.argv[1])); rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one entry per cb_gemm call: its kernel (4-wave gemm_kernel / gemm_dma_kernel or 8-wave gemm8_kernel); the split-K reduce kernel that
# follows an 8-wave launch belongs to the same call (its signature mentions cbgemm::GP: match on the kernel's own name)
g = []
for r in rows:
    name = r["Kernel_Name"]
    if "splitk_reduce_kernel" in name:
        if g:
            g[-1]["End_Timestamp"] = str(int(g[-1]["End_Timestamp"]) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        continue
    if "gemm_kernel" in name or "gemm8_kernel" in name or "gemm_dma_kernel" in name:
        g.append(dict(r))
g = g[-len(calls):]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for c, r in zip(calls, g):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (c["form"], c["conv"], c["M"], c["N"], c["K"], c["split"])
    a = agg[key]; a[0] += d; a[1] += 2.0 * c["M"] * c["N"] * c["K"] * c.get("batch", 1); a[2] += 1
tot = sum(a[0] for a in agg.values())
print(f"total gemm time {tot/1e3:.3f} ms over {len(calls)} launches")
print(f"{'form':6s} {'conv':5s} {'M':>7s} {'N':>6s} {'K':>6s} {'spl':>3s} {'n':>3s} {'us/launch':>9s} {'TF/s':>7s} {'ms tot':>7s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:6s} {str(k[1]):5s} {k[2]:7d} {k[3]:6d} {k[4]:6d} {k[5]:3d} {a[2]:3d} {a[0]/a[2]:9.1f} {a[1]/a[0]/1e6:7.1f} {a[0]/1e3:7.3f}")
