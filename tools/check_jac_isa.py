"""Static check of k_jac_coord's generated code: the destination registers of the inline-asm camera-row requests
(global_load_dwordx4 between #ASMSTART/#ASMEND) must not be read or written by any instruction before the inline-asm
s_waitcnt that follows them -- the compiler does not know those registers are in flight (ba_model_kernels.hip,
jac_issue_cam / jac_wait_cam).  Usage: python tools/check_jac_isa.py [file.s]   (without a file: compiles the kernel
file with hipcc -S).  Exit code 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "bundleadjustment.jl_amd", "csrc", "ba_model_kernels.hip")


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def check(text):
    bad, nkern, nreq = [], 0, 0
    for m in re.finditer(r"^(_ZN\S*k_jac_coord\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        nkern += 1
        inflight = {}   # register -> line of the request
        in_asm = False
        for ln, line in enumerate(body):
            code = line.split(";")[0].strip()
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            if not code or code.endswith(":"):
                continue
            if in_asm and code.startswith("global_load_dwordx4"):
                ops = code.split(None, 1)[1].split(",")
                touched = regs(",".join(ops[1:])) & set(inflight)
                if touched:
                    bad.append((name, ln, code, sorted(touched)))
                for r in regs(ops[0]):
                    inflight[r] = ln
                nreq += 1
                continue
            if in_asm and code.startswith("s_waitcnt"):
                inflight = {}
                continue
            touched = regs(code) & set(inflight)
            if touched:
                bad.append((name, ln, code, sorted(touched)))
    return bad, nkern, nreq


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                                   "-S", "--cuda-device-only", "-o", out, SRC])
            text = open(out).read()
    bad, nkern, nreq = check(text)
    print(f"{nkern} k_jac_coord kernels, {nreq} inline-asm row requests checked, {len(bad)} violations")
    for b in bad[:20]:
        print("  ", b)
    return 1 if bad or nkern == 0 or nreq == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
