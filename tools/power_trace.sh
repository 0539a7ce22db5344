#!/bin/bash
# Sample socket power and sclk every ~0.25 s while a command runs:  bash tools/power_trace.sh out.txt <command ...>
# (rocm-smi takes ~0.2 s per call; the samples are instantaneous readings of the board's averaged power telemetry)
OUT=$1; shift
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | paste - - ; done ) > $OUT &
SP=$!
"$@"
RC=$?
kill $SP 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
pw, ck = [], []
for ln in open(sys.argv[1]):
    m = re.search(r"\((\d+)Mhz\)\s+(\d+(?:\.\d+)?)", ln)
    if m:
        ck.append(int(m.group(1))); pw.append(float(m.group(2)))
if pw:
    n = len(pw)
    busy = [i for i in range(n) if pw[i] > 400]
    print(f"power trace: {n} samples, {len(busy)} under load; mean {sum(pw[i] for i in busy) / max(1, len(busy)):.0f} W, "
          f"share >= 1350 W: {sum(pw[i] >= 1350 for i in busy) / max(1, len(busy)):.2f}, >= 1200 W: {sum(pw[i] >= 1200 for i in busy) / max(1, len(busy)):.2f}, "
          f"< 1000 W: {sum(pw[i] < 1000 for i in busy) / max(1, len(busy)):.2f}; mean sclk under load {sum(ck[i] for i in busy) / max(1, len(busy)):.0f} MHz")
PY
exit $RC
