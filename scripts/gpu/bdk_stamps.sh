#!/bin/bash
# In-kernel stamps of the k-slice GEMM (library built with -DBDK_TIMING as libgpullama_hip_bdkt.so): mean work / barrier-wait cycles per role and class.
set -u
O=${1:-gpurun_out/bdk_stamps}; mkdir -p $O
for P in 2 3; do
  GL3_LIB=$PWD/gpullama3.java_amd/libgpullama_hip_bdkt.so GL3_BDK=1 GL3_BDK_P=$P GL3_BDK_GU=0 GL3_BDK_DA=${BDK_DA:-4} timeout 300 python scripts/bd_only.py qwen3-4b 32 1 > $O/raw_p$P.log 2>&1
  python3 - $O/raw_p$P.log <<'PY'
import re,sys,collections
d=collections.defaultdict(list)
for l in open(sys.argv[1]):
    m=re.match(r'bdk EPI (\d+) P (\d+) rows (\d+) nb (\d+) role (\d+) rounds (\d+): work (\d+) barrier (\d+) total (\d+)',l)
    if m:
        e,p,rows,nb,role,rnd,w,b,t=map(int,m.groups())
        d[(e,p,rows,nb,role,rnd)].append((w,b,t))
for k in sorted(d):
    v=d[k]; n=len(v)
    w=sum(x[0] for x in v)/n; b=sum(x[1] for x in v)/n; t=sum(x[2] for x in v)/n
    print("EPI %d P %d rows %5d nb %3d role %d rounds %3d: n %5d  work/round %6.0f  wait/round %6.0f  total %7.0f" % (*k,n,w/k[5],b/k[5],t))
PY
  tail -1 $O/raw_p$P.log
  rm -f $O/raw_p$P.log
done
