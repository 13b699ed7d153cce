#!/usr/bin/env python3
"""Reads the rocprofv3 kernel trace of tools/gate_probe.py: for the three series runs (with the gate,
without, with it again) how long each step's pipeline kernel and generator kernel took and how much of
the pipeline kernel's time a generator kernel was running beside it.
  python tools/gate_trace.py KERNEL_TRACE.csv STEPS"""
import csv
import json
import sys


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
    steps = int(sys.argv[2])
    et = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'et_stream_kernel' in r['Kernel_Name']]
    syn = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'synth_kernel' in r['Kernel_Name']]
    # the probe launches: 2 warm-up steps, 3 + 3 kernels alone, then 3 series of `steps`
    et_series = et[2 + 3:]
    out = {}
    for k, name in enumerate(('with_gate', 'without_gate', 'with_gate_again')):
        part = et_series[k * steps:(k + 1) * steps]
        if len(part) < steps:
            break
        dur, over, span = [], [], []
        for s, e in part[1:-1]:
            dur.append((e - s) / 1e6)
            o = sum(max(0, min(e, b) - max(s, a)) for a, b in syn)
            over.append(o / (e - s))
        for (s0, e0), (s1, e1) in zip(part[1:-2], part[2:-1]):
            span.append((s1 - s0) / 1e6)
        out[name] = {'et_kernel_ms_mean': sum(dur) / len(dur), 'et_kernel_ms_max': max(dur),
                     'share_of_et_kernel_time_with_a_generator_kernel_running': sum(over) / len(over),
                     'start_to_start_ms_mean': sum(span) / len(span)}
    syn_d = [(e - s) / 1e6 for s, e in syn]
    out['generator_kernel_ms_min_max'] = [min(syn_d), max(syn_d)]
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
