"""Host-side logic that needs no GPU: java.util.Random stream, GGUF round trip, quantisers."""
import os

import numpy as np


def test_java_random_stream(pkg):
    jr = pkg.javarand
    assert jr.JavaRandom(42).next_int() == -1170105035           # well-known new Random(42).nextInt()
    # LlamaBench token stream (J/bench/LlamaBench.java:188-193), KATs from SURVEY.md §8d
    assert jr.bench_tokens(128256, 8) == [16538, 38523, 90864, 87572, 14706, 105637, 41921, 91574]
    assert jr.bench_tokens(151936, 8) == [73242, 98171, 117232, 88212, 73202, 59685, 40129, 133046]
    assert jr.bench_tokens(512, 8) == [372, 27, 349, 24, 158, 482, 141, 362]   # power-of-two branch
    assert all(0 <= t < 1000 for t in jr.bench_tokens(1000, 500))


def test_quantizers_follow_ggml_reference(pkg):
    s = pkg.synth
    w = np.zeros(64, np.float32)
    w[:4] = [1.27, -0.635, 0.005, 0.0149]
    raw = s.quantize_q8_0(w)
    assert raw.size == 68
    d = raw[:2].copy().view(np.float16)[0]
    assert d == np.float16(np.float32(1.27) / np.float32(127))
    assert list(raw[2:6].view(np.int8)) == [127, -64, 1, 1]      # -63.5 -> -64 (half away), 0.5 -> 1, 1.49 -> 1
    assert np.all(raw[34:] == 0)                                  # all-zero block: d = 0, q = 0
    w = np.zeros(32, np.float32)
    w[0], w[1], w[17] = -8.0, 7.0, 3.0
    raw = s.quantize_q4_0(w)
    assert raw[:2].copy().view(np.float16)[0] == 1.0              # d = max/-8 with max = -8
    assert raw[2] & 0xF == 0 and raw[3] & 0xF == 15 and raw[3] >> 4 == 11 and raw[2] >> 4 == 8


def test_gguf_round_trip(pkg, tmp_path):
    s = pkg.synth
    for name, wt in [("tiny-llama", s.GGML_Q8_0), ("tiny-qwen3", s.GGML_Q4_0), ("tiny-llama-tied", s.GGML_F16)]:
        m = s.make_numpy(s.CONFIGS[name], wtype=wt, seed=3)
        path = str(tmp_path / (name + ".gguf"))
        m.write_gguf(path)
        m2 = s.SynthModel.from_gguf(path)
        c1, c2 = m.cfg, m2.cfg
        for f in ("arch", "dim", "hidden", "n_layers", "n_heads", "n_kv_heads", "head_size", "vocab", "ctx", "tied"):
            assert getattr(c1, f) == getattr(c2, f), f
        assert abs(c1.rms_eps - c2.rms_eps) < 1e-12 and c1.rope_theta == c2.rope_theta
        assert set(m.tensors) == set(m2.tensors)
        for k, (raw, ty, rows, cols) in m.tensors.items():
            raw2, ty2, rows2, cols2 = m2.tensors[k]
            assert (ty, rows, cols) == (ty2, rows2, cols2) and np.array_equal(raw, raw2)
        assert m2._gguf.tensor_data_offset % 32 == 0


def test_torch_generator_matches_block_layout(pkg):
    s = pkg.synth
    cfg = s.CONFIGS["tiny-llama"]
    for wt in (s.GGML_Q8_0, s.GGML_Q4_0, s.GGML_F16):
        m = s.make_torch(cfg, wtype=wt, seed=1)
        for k, (raw, ty, rows, cols) in m.tensors.items():
            assert raw.dtype == np.uint8 and raw.size == pkg.gguf.byte_size(ty, rows * cols)
    # the torch and NumPy quantisers implement the same rule
    import torch
    w = torch.randn(64, 64, generator=torch.Generator().manual_seed(0)) * 0.02
    assert np.array_equal(s._t_quantize_q8_0(w).numpy(), s.quantize_q8_0(w.numpy()))
    assert np.array_equal(s._t_quantize_q4_0(w).numpy(), s.quantize_q4_0(w.numpy()))


def test_biased_int_accumulator_trick_is_exact_over_the_whole_block_sum_range():
    """pf_gemm / bd_gemm start the int8 MFMA accumulator at 0x4B400000: reinterpreted as f32 that is 12582912 + isum, and
    subtracting 12582912.0f must give exactly (float)isum for every possible Q8_0 block sum (|isum| <= 32 * 127 * 128)."""
    import numpy as np
    lim = 32 * 127 * 128
    isum = np.arange(-lim, lim + 1, dtype=np.int64)
    bits = (np.int64(0x4B400000) + isum).astype(np.uint32)
    got = bits.view(np.float32) - np.float32(12582912.0)
    assert got.dtype == np.float32 and np.array_equal(got, isum.astype(np.float32))
    assert np.all(np.signbit(got[isum == 0]) == False)      # +0, as (float)0


def test_recorded_bench_lines_follow_the_driver_contract():
    """profiles/r01_bench_*.json are bench.py lines recorded on an MI355X in round 1; the keys the driver and the judge read
    must all be there (metric / value / unit / n_gpus / steps / warmup / ms_per_step / roofline / cpu_baseline ...)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r01_bench_*.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (f, k)
        assert d["unit"] == "tok/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["parity_rel_err_fullsize"] == 0.0
    d = json.load(open(os.path.join(root, "profiles", "r01_bench_8b.json")))
    assert d["metric"] == "tg128 tok/s (llama-bench), Llama-3-8B Q8_0" and d["pp"]["batch"] == 512


def test_bench_gpus_n_never_runs_on_fewer_devices():
    """`bench.py --gpus N` must not silently benchmark fewer GPUs (round-2 review): without a launcher it starts its own N ranks
    and refuses when fewer than N devices are visible; with a launcher whose WORLD_SIZE differs from N it refuses as well.  No GPU
    here, so both refusals can be observed: non-zero exit, the reason on stderr, and no JSON line on stdout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GL3_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0
        assert "--gpus 2 requested but only" in r.stderr
        assert "n_gpus" not in r.stdout
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in r2.stderr
    assert "n_gpus" not in r2.stdout
