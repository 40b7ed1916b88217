#!/bin/bash
# Per-wavefront placement + K-walk cycles of bdw_gemm_kernel (library built with -DBDW_TIMING as libgpullama_hip_bdkt.so): does a wavefront run slower
# when its CU hosts a second one?  (device printf from every wavefront: ~5 minutes of box time)
set -u
O=${1:-gpurun_out/bdw_stamps}; mkdir -p $O
GL3_LIB=$PWD/gpullama3.java_amd/libgpullama_hip_bdkt.so timeout 300 python scripts/bd_only.py qwen3-4b 32 1 > $O/raw.log 2>&1
python3 - $O/raw.log <<'PY'
import re,sys,collections
recs=collections.defaultdict(list)
for l in open(sys.argv[1]):
    m=re.match(r'bdw EPI (\d+) rows (\d+) nb (\d+) wg (\d+) xcc (\d+) se (\d+) cu (\d+) simd (\d+) start (\d+) cycles (\d+)',l)
    if m:
        e,rows,nb,wg,xcc,se,cu,simd,st,cy=map(int,m.groups())
        recs[(e,rows,nb)].append((wg,xcc,se,cu,simd,st,cy))
for k,v in sorted(recs.items()):
    v.sort(key=lambda r:r[5])
    launches=[]; cur=[v[0]]
    for r in v[1:]:
        if r[5]-cur[0][5] > 400: launches.append(cur); cur=[r]
        else: cur.append(r)
    launches.append(cur)
    by=collections.defaultdict(list); ncu=[]
    for L in launches:
        occ=collections.Counter((r[1],r[2],r[3]) for r in L)
        ncu.append(len(occ))
        for r in L: by[occ[(r[1],r[2],r[3])]].append(r[6])
    print("EPI %d rows %d nb %d: %d launches, %d wavefronts in the first, CUs used %s" % (*k,len(launches),len(launches[0]),sorted(set(ncu))))
    for n in sorted(by): print("   wavefronts sharing their CU with %d others of the launch: n %6d  mean %7.0f cycles  (%.0f per tile)" % (n-1,len(by[n]),sum(by[n])/len(by[n]),sum(by[n])/len(by[n])/((k[2]+3)//4)))
PY
tail -1 $O/raw.log; head -3 $O/raw.log; rm -f $O/raw.log
