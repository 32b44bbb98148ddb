"""registers / LDS / kernel-argument bytes of every kernel of one .hip source (device-only compile + code-object notes):
    python tools/kernel_resources.py clipbert_amd/csrc/gemm_inst_bf16_64x64.hip [name filter]"""
import re,sys,subprocess
src=sys.argv[1]
subprocess.run(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-ffp-contract=fast","-I","/root/repo/include","-I","/root/repo/clipbert_amd/csrc","--offload-device-only","-c",src,"-o","/tmp/res/dev.o"],check=True)
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler","--type=o","--targets=hipv4-amdgcn-amd-amdhsa--gfx950","--input=/tmp/res/dev.o","--output=/tmp/res/dev.elf","--unbundle"],check=True)
txt=subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf","--notes","/tmp/res/dev.elf"],capture_output=True,text=True).stdout
recs=[];cur={}
for line in txt.splitlines():
    m=re.match(r'\s+(- )?\.(\w+):\s+(.*)',line)
    if not m: continue
    _d,k,v=m.groups()
    if k=='agpr_count' and cur: recs.append(cur);cur={}
    if k in ('offset','size','value_kind'): continue
    cur[k]=v
recs.append(cur)
flt=sys.argv[2] if len(sys.argv)>2 else ''
for r in recs:
    n=r.get('name','')
    if flt and flt not in n: continue
    print(n.replace('_ZN6cbgemm','')[:150], '| vgpr',r.get('vgpr_count'),'agpr',r.get('agpr_count'),'sgpr',r.get('sgpr_count'),'spill',r.get('vgpr_spill_count'),'lds',r.get('group_segment_fixed_size'),'karg',r.get('kernarg_segment_size'))
