"""Static check of the device code inside libdmn_hip.so: no packed-FP32 VALU instruction may be present.

Why: with ``v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32`` in the kernels (the SLP vectoriser emits them for the
epilogues) this library returned wrong GroupNorm partial sums and linear-attention merge rows in lanes 48..63 of a
wave whenever a kernel of another stream shared the compute unit (measured on MI355X, ROCm 7.2). The build switches
them off (``build.py``); this module makes a dropped flag, a toolchain change or a vector-typed source line fail the
build and the CPU test suite instead of silently bringing the wrong results back.

``python -m diffusion_model_nemo_amd.isa_check [path/to/lib.so]`` prints the per-code-object instruction counts.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from typing import Dict, List

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FORBIDDEN = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
MFMA = re.compile(r"\bv_mfma_")


def _tool(name: str) -> str:
    exe = os.path.join(LLVM_BIN, name)
    if not os.path.exists(exe):
        exe = shutil.which(name) or ""
    if not exe:
        raise RuntimeError(f"{name} not found (looked in {LLVM_BIN} and PATH)")
    return exe


def extract_code_objects(lib_path: str, workdir: str) -> List[str]:
    """Unbundles the gfx950 code objects of a HIP fat binary into `workdir` (the tool writes next to its input, so the
    library is copied there first) and returns their paths."""
    local = os.path.join(workdir, os.path.basename(lib_path))
    shutil.copy(lib_path, local)
    subprocess.run([_tool("llvm-objdump"), "--offloading", local], check=True, capture_output=True, text=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn-amd-amdhsa--gfx950" in f)


def disassemble(code_object: str) -> str:
    r = subprocess.run([_tool("llvm-objdump"), "-d", "--mcpu=gfx950", code_object], check=True, capture_output=True, text=True)
    return r.stdout


def instruction_census(lib_path: str) -> Dict[str, Dict[str, int]]:
    """code object name -> {"forbidden": n, "mfma": n, "kernels": n, "lines": n} for every gfx950 code object of the
    library."""
    out: Dict[str, Dict[str, int]] = {}
    with tempfile.TemporaryDirectory(prefix="dmn_isa_") as tmp:
        objs = extract_code_objects(lib_path, tmp)
        if not objs:
            raise RuntimeError(f"{lib_path} holds no gfx950 code object")
        for co in objs:
            text = disassemble(co)
            lines = text.splitlines()
            out[os.path.basename(co)] = {
                "forbidden": sum(1 for ln in lines if FORBIDDEN.search(ln)),
                "mfma": sum(1 for ln in lines if MFMA.search(ln)),
                "kernels": sum(1 for ln in lines if ln.endswith(">:") and not ln.startswith(" ")),
                "lines": len(lines),
            }
    return out


def assert_no_packed_fp32(lib_path: str) -> Dict[str, Dict[str, int]]:
    census = instruction_census(lib_path)
    bad = {k: v["forbidden"] for k, v in census.items() if v["forbidden"]}
    if bad:
        raise RuntimeError(
            f"{lib_path}: packed-FP32 instructions (v_pk_add/mul/fma_f32) in the device code: {bad}. They gave wrong "
            "sums beside a foreign stream on MI355X; keep `-fno-slp-vectorize -Xclang -target-feature -Xclang "
            "-packed-fp32-ops` in the build and do not use float vector arithmetic in device code.")
    return census


# Kernels whose workgroups may wait for each other inside a launch (the GroupNorm exchange, csrc/dmn_handoff.h): the
# convolution with the fused block tail (last template argument true) and the linear attention's second launch with
# TAIL and several slices per image (TAIL true, ONE false). Two such 256-thread workgroups must fit one CU together, so
# a wave may hold at most half a SIMD's 512 registers (vector + accumulation).
EXCHANGING = re.compile(r"conv_igemm_kernelI\w*ELb[01]ELb1EEEv8ConvArgs|la_fused_out_kernelI\w*ELb1ELb0EEEv16LinAttnFusedArgs")
MAX_EXCHANGING_REGS = 256


def register_census(lib_path: str) -> Dict[str, int]:
    """kernel symbol -> registers per lane (`.vgpr_count` of the code-object metadata: vector + accumulation registers
    as the hardware allocates them) for every gfx950 kernel of the library."""
    out: Dict[str, int] = {}
    with tempfile.TemporaryDirectory(prefix="dmn_isa_") as tmp:
        for co in extract_code_objects(lib_path, tmp):
            r = subprocess.run([_tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True)
            name = None
            for ln in r.stdout.splitlines():
                m = re.match(r"\s+(?:- )?\.name:\s+(\S+)", ln)
                if m:
                    name = m.group(1)
                m = re.match(r"\s+(?:- )?\.vgpr_count:\s+(\d+)", ln)
                if m and name:
                    out[name] = int(m.group(1))
                    name = None
    return out


def assert_exchanging_kernels_fit_half_a_cu(lib_path: str) -> Dict[str, int]:
    regs = {k: v for k, v in register_census(lib_path).items() if EXCHANGING.search(k)}
    if not regs:
        raise RuntimeError(f"{lib_path}: no exchanging kernel found (symbol pattern out of date?)")
    bad = {k: v for k, v in regs.items() if v > MAX_EXCHANGING_REGS}
    if bad:
        raise RuntimeError(
            f"{lib_path}: kernels whose workgroups wait for each other use more than {MAX_EXCHANGING_REGS} registers: {bad}. "
            "Two of their workgroups no longer fit one CU, and two launch chains can starve each other's waits; keep "
            "their `__launch_bounds__(256, 2)`.")
    return regs


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "lib", "libdmn_hip.so")
    total = {"forbidden": 0, "mfma": 0, "kernels": 0}
    for name, c in assert_no_packed_fp32(path).items():
        print(f"{name}: {c['kernels']} symbols, {c['mfma']} v_mfma_*, {c['forbidden']} v_pk_(add|mul|fma)_f32")
        for k in total:
            total[k] += c[k]
    print(f"total: {total}")
    regs = assert_exchanging_kernels_fit_half_a_cu(path)
    print(f"exchanging kernels: {len(regs)}, registers per lane {min(regs.values())}..{max(regs.values())} (limit {MAX_EXCHANGING_REGS})")
