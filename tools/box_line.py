"""Dev tool (GPU box): one line per box for profiles/<round>/box_spread.txt - `python bench.py --no-cpu-baseline` reduced to the launch time of the main leg,
the steady leg with its on-box power / clock, the opt-in int8 kernel, the fp32 kernel and the two end-to-end legs (feature containers: whole command; raw
containers: whole command and steady state).  Run it on several fresh boxes back to back."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
r = d["roofline"]; s = r["power"].get("steady") or {}; x = d["extras"]; st = r.get("steady") or {}
e, er = x.get("e2e") or {}, x.get("e2e_raw") or {}
print("box %s timed %.4f ms frac %.4f | steady %.4f ms frac %.4f, %s W %s MHz %.1f uJ/window | f16i8 %.4f | f32 %.4f (%.3f) | e2e packed %.2f s (BED %s) | e2e raw %.2f s, steady %.3g /s, waits feeders %.1f s device %.1f s of %.1f" % (
    s.get("pci_bus_id"), r["avg_launch_ms"], r["frac"], st.get("avg_launch_ms", float("nan")), st.get("frac", float("nan")), (s.get("socket_power_w") or {}).get("median"),
    (s.get("sclk_mhz") or {}).get("median"), s.get("microjoules_per_window") or float("nan"), x["opt_in_precision"]["avg_launch_ms"], x["other_precision"]["avg_launch_ms"],
    x["other_precision"]["frac"], e.get("wall_s", float("nan")), e.get("bed_matches_expected"), er.get("wall_s", float("nan")), er.get("base_positions_per_s_steady_state") or float("nan"),
    er.get("waiting_for_feeders_s") or 0.0, er.get("waiting_for_device_s") or 0.0, er.get("detect_wall_s") or float("nan")))
