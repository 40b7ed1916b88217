"""Writes the synthetic Llama-3-8B Q8_0 GGUF to /tmp and runs the native C++ host (tools/gl3_bench) on it."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
cfg = pkg.synth.CONFIGS[name]
cfg = pkg.synth.ModelConfig(**{**cfg.__dict__, "ctx": int(os.environ.get("GL3_CTX", "648"))})
t0 = time.time()
m = pkg.synth.make_torch(cfg, seed=42, device="cuda")
path = "/tmp/%s.Q8_0.gguf" % name
m.write_gguf(path)
print("wrote %s (%.1f GB) in %.1f s" % (path, os.path.getsize(path) / 1e9, time.time() - t0), flush=True)
del m
torch.cuda.empty_cache()
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gl3_bench")
subprocess.run([exe, "-m", path] + (sys.argv[2:] or ["-p", "512", "-n", "128", "-b", "512", "-r", "3"]), check=False)
os.remove(path)
