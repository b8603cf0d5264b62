"""Builds libdmn_hip.so (hand-written HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

``python -m diffusion_model_nemo_amd.build`` or ``build()``; cross-compiles without a GPU.
Objects are cached under ``csrc/_obj`` and rebuilt when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdmn_hip.so")
SOURCES = ["dmn_conv.hip", "dmn_elem.hip", "dmn_attn.hip", "dmn_linattn.hip", "dmn_sampler.hip", "dmn_unet.cpp", "dmn_api.cpp",
           "dmn_train_gemm.hip", "dmn_train_ops.hip", "dmn_train.cpp", "dmn_peaks.hip"]
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP kernels cannot be built")
    return exe


def _newest_header():
    inc = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc.append(os.path.join(os.path.dirname(HERE), "include", "dmn.h"))
    return max(os.path.getmtime(p) for p in inc)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    obj_dir = os.path.join(CSRC, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    hdr_time = _newest_header()
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
             "-ffp-contract=off", "-fno-honor-nans",
             # No packed-FP32 instructions (v_pk_add/mul/fma_f32) in device code, neither from the SLP vectoriser nor from
             # vector-typed source: with them, kernels of this library returned wrong sums in lanes 48..63 of a wave
             # whenever another stream's kernel shared the CU (measured on MI355X with ROCm 7.2: GroupNorm partial sums
             # short by a lane's share, linear-attention merge rows off by up to 20 %; never when the kernels ran alone).
             # The scalar forms cost nothing measurable here (the denoising step got 0.5 % faster). The host pass prints
             # "not a recognized feature" for the second option and ignores it.
             "-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    # Kernels whose MFMA accumulators feed VALU code inside the loop (softmax, operand re-packing): MFMA results in
    # plain VGPRs, or the compiler shuttles them through the accumulator file and back on every iteration.
    extra = {"dmn_linattn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "dmn_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
             # hardware float atomics for the split reductions of the training kernels
             "dmn_train_gemm.hip": ["-munsafe-fp-atomics"], "dmn_train_ops.hip": ["-munsafe-fp-atomics"]}
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_time):
            cmd = [hipcc, *flags, *extra.get(src, []), "-x", "hip", "-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB_PATH):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs])
        # the flags above are only as good as what they produce: disassemble the result and refuse a library that
        # holds a packed-FP32 instruction after all (toolchain change, vector-typed source line, dropped flag)
        # ... or whose exchanging kernels outgrew half a SIMD's registers (two of their workgroups must fit one CU)
        from .isa_check import assert_exchanging_kernels_fit_half_a_cu, assert_no_packed_fp32
        try:
            assert_no_packed_fp32(LIB_PATH)
            assert_exchanging_kernels_fit_half_a_cu(LIB_PATH)
        except Exception:
            os.replace(LIB_PATH, LIB_PATH + ".rejected")   # never leave a library behind that must not be loaded
            raise
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
