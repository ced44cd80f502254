"""Runs bench.py over the BASELINE configs / mask strides and prints one summary line each."""
import json
import subprocess
import sys

RUNS = [("h36m_351", 128, 5), ("h36m_351", 128, 10), ("h36m_351", 128, 20),
        ("h36m_81", 256, 4), ("h36m_81", 256, 10), ("h36m_81", 256, 20)]
for cfg, batch, ms in RUNS:
    out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--batch", str(batch), "--mask-stride", str(ms),
                          "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(cfg, ms, "FAILED", out.stderr[-500:])
        continue
    d = json.loads(line[-1])
    k = d["kernel_ms_per_forward"]
    print(f"{cfg} batch {batch} s_in {ms}: {d['value']:.0f} seq/s  {d['ms_per_step']:.3f} ms/step  model {d['roofline']['model_tflops']} TF  "
          f"dominant {d['roofline']['kernel']} {d['roofline']['achieved']} TF  spatial {k.get('spatial_stack')} ms  compact {k.get('compact_frames')} ms", flush=True)
