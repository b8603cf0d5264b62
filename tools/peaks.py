"""Prints the measured ceilings of the GPU (diffusion_model_nemo_amd.peaks.measure) and optionally saves them.

python tools/peaks.py [--out gpurun_out/peaks.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diffusion_model_nemo_amd.peaks import measure  # noqa: E402


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = measure()
    for s in r["stream"]:
        print(f"{s['kind']:5s} {s['bytes'] / 1e6:9.1f} MB  {s['launch_us']:9.1f} us  {s['GBs']:8.0f} GB/s")
    for m in r["mfma"]:
        print(f"{m['instruction']:26s} {m['waves_per_simd']} waves/SIMD  {m['launch_ms']:7.3f} ms  {m['TFLOPs']:8.1f} TFLOP/s")
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(r, fh, indent=1)
