import re,collections,sys
acc=collections.defaultdict(lambda: [0,0,0,0])
for l in open(sys.argv[1]):
    m=re.search(r"\[(.*?)\].*g3t NFR (\d) WCN (\d) J \d+ wave \d+ stages (\d+): barrier (\d+) scale (\d+) total (\d+)",l)
    if not m: 
        m2=re.search(r"g3t NFR (\d) WCN (\d) J \d+ wave \d+ stages (\d+): barrier (\d+) scale (\d+) total (\d+)",l)
        if not m2: continue
        k=(m2.group(1),m2.group(2),m2.group(3)); g=m2.groups()[3:]
    else:
        k=(m.group(2),m.group(3),m.group(4)); g=m.groups()[4:]
    a=acc[k]; a[0]+=1
    for i in range(3): a[i+1]+=int(g[i])
for k,a in sorted(acc.items()):
    n=a[0]; st=int(k[2]); print("NFR %s WCN %s stages %s"%k, "n",n, "per stage: barrier %.0f scale %.0f total %.0f"%(a[1]/n/st,a[2]/n/st,a[3]/n/st))
