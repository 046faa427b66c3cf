"""Is the headline loop bound by the device chain or by the host's enqueue rate?  Runs bench.py (child processes) with and
without a device stall in front of the timed loop (--stall_ms: the host then enqueues the whole loop ahead of the device) at two
loop lengths; the SLOPE between the lengths is the step period without the stall's constant.  Equal slopes: device-bound.
Usage: python tools/period_probe.py [extra bench.py args]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(stall, steps, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "20", "--no_cpu_baseline", "--no_e2e",
           "--no_kernel_timing", "--stall_ms", str(stall)] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    return d["ms_per_step"] * steps, d.get("host_enqueue_ms_per_step")


if __name__ == "__main__":
    extra = sys.argv[1:]
    for rep in range(2):
        for stall in (0, 150):
            (a, ea), (b, eb) = run(stall, 400, extra), run(stall, 1600, extra)
            print("rep %d stall %3d ms: period %.4f ms/step (loop 400: %.1f ms, loop 1600: %.1f ms; host enqueue %.4f ms/step)"
                  % (rep, stall, (b - a) / 1200, a, b, eb), flush=True)
