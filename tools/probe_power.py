"""Dev tool (GPU box): what the box offers for sampling socket power and shader clock without a subprocess - sysfs hwmon / pp_dpm_sclk /
gpu_metrics - and what one read costs.  Used once to choose the sources of deepmod_amd/powerlog.py."""
import glob, os, time

for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    dev = os.path.join(card, "device")
    if not os.path.exists(os.path.join(dev, "vendor")):
        continue
    print(card, open(os.path.join(dev, "vendor")).read().strip())
    for h in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
        for f in sorted(os.listdir(h)):
            p = os.path.join(h, f)
            if os.path.isfile(p) and (f.startswith(("power", "freq", "temp1")) or f == "name"):
                try:
                    t0 = time.perf_counter()
                    v = open(p).read().strip()
                    dt = time.perf_counter() - t0
                    print("   %-28s %-24s %.2f ms" % (f, v[:24], dt * 1e3))
                except OSError as e:
                    print("   %-28s %r" % (f, e))
    for f in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "current_compute_partition"):
        p = os.path.join(dev, f)
        if os.path.exists(p):
            try:
                t0 = time.perf_counter()
                v = open(p).read().strip().replace("\n", " | ")
                print("   %-28s %-60s %.2f ms" % (f, v[:60], (time.perf_counter() - t0) * 1e3))
            except OSError as e:
                print("   %-28s %r" % (f, e))
    p = os.path.join(dev, "gpu_metrics")
    if os.path.exists(p):
        t0 = time.perf_counter()
        b = open(p, "rb").read()
        print("   gpu_metrics %d bytes, header %s, %.2f ms" % (len(b), b[:4].hex(), (time.perf_counter() - t0) * 1e3))
