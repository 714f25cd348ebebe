# usage: res.sh file.hip [filter]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -load-store-opt -mllvm -amdgpu-atomic-optimizer-strategy=None -I../include -Icsrc -Rpass-analysis=kernel-resource-usage -c $1 -o /dev/null 2>&1 | python3 -c "
import sys,re
cur=None;d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}
    for k in ('VGPRs','AGPRs','SGPRs Spill','VGPRs Spill','Occupancy \[waves/SIMD\]','ScratchSize \[bytes/lane\]'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: d[cur][k.replace('\\\\','')]=m.group(1)
import subprocess
for k,v in d.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    name=name.replace('uc::(anonymous namespace)::','').split('(')[0]
    print('%-46s'%name,' '.join('%s=%s'%(a.split(' [')[0],b) for a,b in v.items()))
"
