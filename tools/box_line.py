"""Dev tool (GPU box): one line per box for profiles/<round>/box_spread.txt - `python bench.py --no-e2e --no-cpu-baseline` reduced to the launch time of the
main leg, the on-box power / clock of the steady leg, extras.shape_ab, the opt-in int8 kernel and the fp32 kernel.  Run it on several fresh boxes back to back."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-e2e", "--no-cpu-baseline"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
r = d["roofline"]; s = r["power"].get("steady") or {}; x = d["extras"]
print("box %s launch %.4f ms frac %.4f | steady %s W %s MHz %.1f uJ/window | 32x32 %.4f ratio %.3f | f16i8 %.4f | f32 %.4f" % (
    s.get("pci_bus_id"), d["ms_per_step"], r["frac"], (s.get("socket_power_w") or {}).get("median"), (s.get("sclk_mhz") or {}).get("median"),
    s.get("microjoules_per_window") or float("nan"), x["shape_ab"]["median_ms_32x32x16"], x["shape_ab"]["ratio_16_over_32"],
    x["opt_in_precision"]["avg_launch_ms"], x["other_precision"]["avg_launch_ms"]))
