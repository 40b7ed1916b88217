"""Wall time of tools/gl3_run invocations on a tiny GGUF (process start + load + loop): python scripts/run_host_timing.py"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
m = pkg.synth.make_numpy(pkg.synth.CONFIGS["tiny-llama"], seed=31)
m.write_gguf("/tmp/t.gguf")
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gl3_run")
for extra in (["-n", "24", "--bos", "1"], ["-n", "24", "--bos", "1", "-b", "4"], ["-n", "24", "--bos", "1", "--temperature", "0.8"]):
    t = time.time()
    out = subprocess.run([exe, "-m", "/tmp/t.gguf", "--ids", "1,2,3,4,5"] + extra, capture_output=True, text=True)
    print("%.1f s" % (time.time() - t), extra, out.stdout.strip()[:60], out.stderr.strip()[-160:])
