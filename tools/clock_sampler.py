"""Samples the GPU's clocks and power while something else runs (rocm-smi once a second; a process of its own, no HIP):
    python tools/clock_sampler.py <out.jsonl> <seconds>
One JSON object per sample: t (s since start), sclk / mclk (MHz), power (W), temperature, perf level -- whatever this rocm-smi reports."""
import json
import subprocess
import sys
import time


def main():
    out, secs = sys.argv[1], float(sys.argv[2])
    t0 = time.time()
    with open(out, "w") as fh:
        while time.time() - t0 < secs:
            rec = {"t": round(time.time() - t0, 1)}
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
                j = json.loads(r.stdout)
                card = j.get("card0", {})
                for k, v in card.items():
                    kl = k.lower()
                    if "sclk" in kl or "mclk" in kl or "power" in kl or "performance" in kl or ("temperature" in kl and "junction" in kl):
                        rec[k] = v
            except Exception as e:
                rec["err"] = f"{type(e).__name__}: {e}"[:200]
            fh.write(json.dumps(rec) + "\n")
            fh.flush()
            time.sleep(1.0)


if __name__ == "__main__":
    main()
