#!/usr/bin/env python3
"""Scans the gfx950 code of libsvs_hip.so for instruction forms that must not be in it.

Today one rule: no packed-fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) with an `op_sel:[...]` modifier that
takes the HIGH half of a source for the low result.  On gfx950 such an instruction returns garbage while a bf16 MFMA
(v_mfma_f32_16x16x32_bf16) of any other wave is executing on the same CU (tools/attic/stress_victims.py); the
compiler produces the form when it SLP-vectorises complex arithmetic, which is why stft.hip / mrstft.hip are built with
-fno-slp-vectorize (svs_unet_pytorch_amd/build.py).

    python tools/check_isa.py [path/to/libsvs_hip.so]      exit code 1 and a listing if a forbidden form is found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FORBIDDEN = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[")


def device_code_objects(lib, workdir):
    """Every gfx950 code object embedded in the library's .hip_fatbin section (one bundle per translation unit)."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        piece = os.path.join(workdir, f"bundle{i}.bin")
        open(piece, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, f"dev{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={piece}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def scan(lib):
    hits, n_inst, n_kernels = [], 0, 0
    with tempfile.TemporaryDirectory() as wd:
        for co in device_code_objects(lib, wd):
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
            sym = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    sym = m.group(1); n_kernels += 1
                    continue
                if "\t" in line or "  v_" in line or "  s_" in line:
                    n_inst += 1
                    if FORBIDDEN.search(line):
                        hits.append((sym, line.strip()))
    return hits, n_inst, n_kernels


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "svs_unet_pytorch_amd", "libsvs_hip.so")
    hits, n_inst, n_kernels = scan(lib)
    print(f"{lib}: {n_kernels} symbols, {n_inst} instructions scanned, {len(hits)} forbidden packed-fp32 op_sel forms")
    for sym, line in hits[:20]:
        print("  ", sym[:60], "|", line[:120])
    sys.exit(1 if hits else 0)
