import sys, time
sys.path.insert(0, "/root/repo")
import bench
t=time.time(); print(bench.probe_peaks(0), time.time()-t)
