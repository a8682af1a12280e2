#!/bin/bash
# Sample the GPU's clocks and power (rocm-smi) every ~0.2 s while a command runs; prints the samples then the command's output.
#   bash scripts/smi_sample.sh <label> <command...>
label=$1; shift
log=/tmp/smi_$$.txt; : > $log
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' >> $log; echo >> $log; sleep 0.2; done ) &
spid=$!
"$@" > /tmp/smi_cmd_$$.txt 2>&1
kill $spid 2>/dev/null; wait $spid 2>/dev/null
echo "== $label"
python3 - $log <<'PY'
import re, sys
sc, pw = [], []
for ln in open(sys.argv[1]):
    m = re.search(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", ln) or re.search(r"sclk[^0-9]*(\d+)\s*Mhz", ln, re.I)
    if m: sc.append(int(m.group(1)))
    m = re.search(r"Power \(W\):\s*([0-9.]+)", ln)
    if m: pw.append(float(m.group(1)))
def st(v): return f"n={len(v)} min={min(v):.0f} median={sorted(v)[len(v)//2]:.0f} max={max(v):.0f}" if v else "no samples"
print("   sclk MHz:", st(sc)); print("   power W :", st(pw))
PY
head -3 $log | cut -c1-300
grep -E "sl= 1|tile" /tmp/smi_cmd_$$.txt | tail -2 | cut -c1-200
