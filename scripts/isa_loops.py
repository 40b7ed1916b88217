"""Loops of one kernel in a built object, read from the disassembly (no GPU): for every backward branch the number of instructions, VALU
adds / multiplies, MFMAs, LDS reads and s_waitcnt lgkmcnt(0) in the body.  A dependent chain whose body shows its ds_reads right in
front of the first use (lgkmcnt(0) a few instructions after the read) waits a full LDS round trip per trip — the pattern behind the
pinned read rings of r4 (profiles/r04_decode_budget.md).

    python scripts/isa_loops.py gpullama3.java_amd/csrc/gl3_api.o attn_head_kernelILi128E            # table
    python scripts/isa_loops.py gpullama3.java_amd/csrc/gl3_api.o moe_router_kernel --show 48        # + bodies with 48 v_add_f32
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        subprocess.check_call(["cp", obj, tmp])
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        dev = [f for f in os.listdir(d) if "amdgcn" in f]
        if not dev:
            sys.exit("no device code object in " + obj)
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(d, dev[0])], text=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--show", type=int, default=-1, help="print the bodies of the loops with exactly this many v_add_f32")
    args = ap.parse_args()
    lines, on = [], False
    for l in device_disassembly(args.obj).split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            if on:
                break
            on = args.kernel in m.group(1) and "$" not in m.group(1)
            if on:
                print(m.group(1))
            continue
        if on and l.strip():
            lines.append(l.split("//")[0].rstrip())
    print("%6s %5s %5s %5s %5s %8s %8s %s" % ("end", "len", "add", "mul", "mfma", "ds_read", "wait(0)", "waits"))
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\d+)", l)
        if not m or int(m.group(1)) < 32768:
            continue
        n = 65536 - int(m.group(1))
        body = lines[max(0, i - n):i + 1]
        cnt = lambda pat: sum(1 for x in body if re.search(pat, x))
        waits = [re.search(r"lgkmcnt\((\d+)\)", x).group(1) for x in body if "lgkmcnt" in x]
        print("%6d %5d %5d %5d %5d %8d %8d %s" % (i, n, cnt("v_add_f32"), cnt(r"v_(pk_)?mul_f32"), cnt("v_mfma"), cnt("ds_read"), waits.count("0"),
                                                 ",".join(waits[:12])))
        if args.show >= 0 and cnt("v_add_f32") == args.show:
            print("\n".join("        " + x.strip() for x in body))


if __name__ == "__main__":
    main()
