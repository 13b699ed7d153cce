#!/usr/bin/env python3
"""Duration of the dominant kernel over the TIMED dispatches of a profiled bench.py run.

    python tools/timed_dispatches.py KERNEL_TRACE.csv WARMUP STEPS [SYMBOL]

rocprofv3's kernel_stats.csv averages every call of a symbol. A bench.py run launches the
pipeline kernel WARMUP times, then STEPS timed times, then again behind the timed region (the
sensor leg's extra steps, the parity legs). This reads the kernel trace, orders the symbol's
dispatches by start time and reports the timed ones [WARMUP, WARMUP + STEPS) on their own --
average, min, max, the idle gap in front of each group -- and names what it excluded and what
those looked like (round 4: dispatches 45-46, the first two behind the timed region, 21.4 and
27.4 ms behind a 3.4 ms idle gap, sat in the quoted 20.18 ms average)."""
import csv
import json
import sys


def main():
    path, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    symbol = sys.argv[4] if len(sys.argv) > 4 else 'et_stream_kernel'
    rows = [r for r in csv.DictReader(open(path)) if symbol in r['Kernel_Name']]
    # the instance with the most calls (a run may also launch other instances of the template once)
    names = {}
    for r in rows:
        names[r['Kernel_Name']] = names.get(r['Kernel_Name'], 0) + 1
    name = max(names, key=names.get)
    rows = sorted((r for r in rows if r['Kernel_Name'] == name), key=lambda r: int(r['Start_Timestamp']))
    t = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    dur = [(e - s) / 1e6 for s, e in t]
    gap = [0.0] + [(t[i][0] - t[i - 1][1]) / 1e6 for i in range(1, len(t))]

    def stat(lo, hi):
        d = dur[lo:hi]
        if not d:
            return None
        return {'dispatches': [lo, hi], 'calls': len(d), 'avg_ms': sum(d) / len(d), 'min_ms': min(d), 'max_ms': max(d),
                'idle_gap_in_front_ms': gap[lo], 'largest_gap_inside_ms': max(gap[lo + 1:hi] or [0.0])}
    out = {'kernel': name, 'calls_total': len(dur), 'all_calls_avg_ms': sum(dur) / len(dur),
           'timed': stat(warm, warm + steps), 'excluded': {'warmup': stat(0, warm), 'behind_the_timed_region': stat(warm + steps, len(dur))},
           'slowest': sorted(({'dispatch': i, 'ms': d, 'idle_gap_in_front_ms': gap[i]} for i, d in enumerate(dur)),
                             key=lambda x: -x['ms'])[:4]}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
