import csv, collections, re, sys
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0,0])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"gemm_kernelIDF16bLi(\d+)ELi(\d+)E(.*)", n)
    if m:
        rest = m.group(3)
        def ld(t):
            return t
        a = "Aroku" 
        names = re.findall(r"(RowkFast|KrowFast|RowkLoader|KrowLoader)IDF16bLi\d+EL[bi](\d)", rest)
        if "S2_" in rest and len(names) == 1: names = names*2
        tag = " x ".join(f"{k}{v}" for k,v in names)
        return f"gemm {m.group(1)}x{m.group(2)} {tag}"
    m = re.search(r"gemm_kernel<.*", n)
    if m: return "gemm(demangled) " + n[-90:]
    return n[:60]
for r in rows:
    key = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][0] += d; agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print("total ms/step", round(tot/1e6/nsteps,3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv)>3 else 24]:
    print(f"{v[0]/1e6/nsteps:7.3f} ms/step n/step={v[1]/nsteps:6.1f} avg={v[0]/v[1]/1e3:7.1f}us  {k}")
