"""Runs the bare MFMA loops of tools/ubench (prebuilt into tools/ubench/bin by hipcc --offload-arch=gfx950) and writes what
they reach on THIS box as JSON: bench.py's `measured_mfma_ceiling_tflops` reads the committed copy (profiles/r0N_ubench_ceilings.json).
    python tools/ubench_ceilings.py <out.json>          measurement tool, not part of the product path"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "tools", "ubench", "bin")
out = {}
r = subprocess.run([os.path.join(B, "mfma_peak")], capture_output=True, text=True, timeout=120)
v = [float(x) for x in re.findall(r"([\d.]+) TFLOP/s", r.stdout)]
out["mfma_f32_tflops"] = max(v)
out["mfma_f32_lines"] = r.stdout.strip().splitlines()[-4:]
r = subprocess.run([os.path.join(B, "bf16x9")], capture_output=True, text=True, timeout=120)
v = [float(x) for x in re.findall(r"-> ([\d.]+) TFLOP/s fp32-equivalent", r.stdout)]
out["bf16x9_fp32_equivalent_tflops"] = max(v)
out["bf16x9_lines"] = [l for l in r.stdout.strip().splitlines() if "TFLOP/s fp32-equivalent" in l]
out["what"] = ("best of the pure-MFMA loops: v_mfma_f32_32x32x2_f32 (fp32 peak 157.3 spec) and the nine-product bf16 inner loop "
               "(v_mfma_f32_32x32x16_bf16 x 9 per fp32 product; x 9 / 8 for the eight-product default)")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("tflops")}))
