#!/bin/bash
# Run on the GPU box (via gpurun): sample socket power and shader clock (rocm-smi, ~4 Hz) while a command runs.
# Usage: bash tools/power_trace.sh <out.txt> <command...>
OUT=$1; shift
( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | grep '^card0'; sleep 0.2; done ) > $OUT.raw 2>&1 &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
python3 - "$OUT" "$*" <<'PY'
import sys
out, cmd = sys.argv[1], sys.argv[2]
pw, sclk = [], []
for l in open(out + ".raw"):
    f = l.strip().split(',')
    if len(f) < 10:
        continue
    try:
        sclk.append(int(f[5].strip('()').replace('Mhz', '')))
        pw.append(float(f[9]))
    except ValueError:
        pass
busy = [(p, c) for p, c in zip(pw, sclk) if p > 0.5 * max(pw)] if pw else []
med = lambda v: sorted(v)[len(v) // 2] if v else None
open(out, "w").write("command: %s\nsamples %d (busy %d)\nsocket power W: idle-ish min %s, busy median %s, max %s (cap 1400 W)\nsclk MHz while busy: median %s, min %s, max %s\n"
                     % (cmd, len(pw), len(busy), min(pw) if pw else None, med([b[0] for b in busy]), max(pw) if pw else None,
                        med([b[1] for b in busy]), min([b[1] for b in busy]) if busy else None, max([b[1] for b in busy]) if busy else None))
print(open(out).read())
PY
rm -f $OUT.raw
exit $RC
