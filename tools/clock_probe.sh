#!/bin/bash
# Shader clock and package power of the device while the headline step runs (read-only rocm-smi
# queries next to a bench run): is the float64 kernel's ~1.75 GHz (profiles/r03_sq_counters_*)
# the power cap? usage: tools/clock_probe.sh OUTDIR [bench args]
out=$1; shift
mkdir -p "$out"
rocm-smi --showclocks --showpower --showmaxpower --showperflevel > "$out/idle.txt" 2>&1
python bench.py --steps 600 --no-configs --no-parity --no-cpu-baseline --no-plain "$@" > "$out/bench.json" 2> "$out/bench.err" &
pid=$!
sleep 12
for i in $(seq 1 40); do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showclocks --showpower --json >> "$out/samples.jsonl" 2>/dev/null
    echo >> "$out/samples.jsonl"
    sleep 0.4
done
wait $pid
python - "$out" <<'PY'
import json, sys, re
out = sys.argv[1]
sclk, power = [], []
for line in open(out + '/samples.jsonl'):
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    for card, f in d.items():
        for k, v in f.items():
            m = re.search(r'\((\d+)Mhz\)', str(v))
            if 'sclk' in k and m:
                sclk.append(int(m.group(1)))
            if 'Power' in k and 'Socket' in k or 'Average Graphics Package Power' in k:
                try:
                    power.append(float(v))
                except ValueError:
                    pass
print('samples', len(sclk), 'sclk MHz min/mean/max', min(sclk or [0]), sum(sclk) / max(len(sclk), 1), max(sclk or [0]))
print('power W min/mean/max', min(power or [0]), sum(power) / max(len(power), 1), max(power or [0]))
PY
