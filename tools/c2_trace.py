#!/usr/bin/env python3
"""configs[1] under `rocprofv3 --kernel-trace`: the 1200 x 1200 launches of tools/c2bench.py split into
what the KERNEL takes and what lies between two launches (dispatch + completion, the part no kernel
change reaches). Launches shorter than 80 us are the single-tile ones.
  python tools/c2_trace.py KERNEL_TRACE.csv"""
import csv
import json
import sys


def q(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))]


def main():
    rows = sorted((r for r in csv.DictReader(open(sys.argv[1])) if 'et_stream_kernel' in r['Kernel_Name']),
                  key=lambda r: int(r['Start_Timestamp']))
    t = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    dur = [(e - s) / 1e3 for s, e in t]
    single = [i for i, d in enumerate(dur) if d < 80]
    d1 = [dur[i] for i in single]
    # back-to-back single-tile launches: start-to-start distances below 80 us
    period = [(t[j][0] - t[i][0]) / 1e3 for i, j in zip(single, single[1:]) if j == i + 1 and (t[j][0] - t[i][0]) / 1e3 < 80]
    gap = [(t[j][0] - t[i][1]) / 1e3 for i, j in zip(single, single[1:]) if j == i + 1 and (t[j][0] - t[i][0]) / 1e3 < 80]
    print(json.dumps({'single_tile_launches': len(d1),
                      'kernel_us': {'min': min(d1), 'p10': q(d1, 0.1), 'median': q(d1, 0.5), 'p90': q(d1, 0.9)},
                      'start_to_start_us_back_to_back': {'p10': q(period, 0.1), 'median': q(period, 0.5)},
                      'end_to_next_start_us': {'p10': q(gap, 0.1), 'median': q(gap, 0.5)},
                      'note': 'with and without diagnostics mixed (tools/c2bench.py issues both); under the profiler'}))


if __name__ == '__main__':
    main()
