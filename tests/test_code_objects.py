"""The gfx950 code objects inside libba_hip.so: no kernel may use scratch memory or a dynamic stack.  Either means that a
device function was NOT inlined into its kernel (a real call: worst-case register budget, spills, one wave per SIMD) or that
an array went to memory -- in round 4 the row-split panel kernels ran 1.5x slower for exactly that reason and nothing but a
kernel trace showed it.  CPU only: reads the metadata notes of the embedded code objects."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bundleadjustment.jl_amd", "libba_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """every gfx950 ELF of every offload bundle in the .hip_fatbin section"""
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf of the ROCm toolchain not found")
def test_no_kernel_uses_scratch_or_a_dynamic_stack(ba, tmp_path):
    assert os.path.exists(LIB)
    sec = str(tmp_path / "fatbin.bin")
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", LIB, sec], check=True)
    blob = open(sec, "rb").read()
    objs = _code_objects(blob)
    assert len(objs) >= 4, "one code object per HIP translation unit that holds kernels"
    kernels, bad = 0, []
    for i, co in enumerate(objs):
        f = str(tmp_path / f"co{i}.elf")
        open(f, "wb").write(co)
        notes = subprocess.run([READELF, "--notes", f], capture_output=True, text=True, check=True).stdout
        # one YAML-like record per kernel; fields in alphabetical order inside a record
        for rec in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", rec)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", rec)
            dyn = re.search(r"\.uses_dynamic_stack:\s+(\w+)", rec)
            if not name or not scratch:
                continue
            kernels += 1
            if int(scratch.group(1)) != 0 or (dyn and dyn.group(1) == "true"):
                bad.append((name.group(1), int(scratch.group(1)), dyn.group(1) if dyn else None))
        # a call instruction anywhere in the code object: a device function that stayed a function (even without scratch it
        # runs on the worst-case register budget)
        dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f], capture_output=True, text=True, check=True).stdout
        ncalls = dis.count("s_swappc_b64")
        if ncalls:
            bad.append((f"code object {i}", f"{ncalls} call instruction(s)", None))
    print(f"{len(objs)} code objects, {kernels} kernels")
    assert kernels >= 100
    assert not bad, f"kernels with scratch / dynamic stack (a device function that was not inlined?): {bad}"
