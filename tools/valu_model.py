#!/usr/bin/env python3
"""How much of a step the vector pipe's ISSUE slots account for: a wave64 VALU instruction occupies
its SIMD for 4 cycles (16 lanes per cycle; float64 and packed float32 alike on gfx950), so a kernel
of I vector instructions EXECUTED per wave-iteration of P pixels per lane (SQ_INSTS_VALU / wave-iterations:
tools/run_sq_counters.sh -- the loop body's static count includes branches a wave does not take) cannot take fewer than
pixels / (64 P) x I x 4 / (4 SIMDs x CUs) shader cycles. Against the measured cycles of the step
(kernel_ms x the shader clock read under load, both from a bench line) that is the share of the
step the vector pipe is issuing -- what is left is memory time not hidden behind it.
  tools/valu_model.py BENCH_LINE.json VALU_PER_ITERATION PIXELS_PER_LANE [CUS]"""
import json
import sys


def main():
    line = json.load(open(sys.argv[1]))
    valu, per_lane = float(sys.argv[2]), int(sys.argv[3])
    cus = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    roof = line['roofline']
    pixels, ms, mhz = roof['pixels_per_launch'], roof['kernel_ms'], roof['sclk_mhz']
    issue = pixels / (64.0 * per_lane) * valu * 4.0 / (4 * cus)
    measured = ms * mhz * 1e3
    print(json.dumps({'kernel': roof['kernel'], 'dtype': line['dtype'], 'pixels': pixels, 'kernel_ms': ms, 'sclk_mhz': mhz,
                      'valu_per_wave_iteration': valu, 'pixels_per_lane': per_lane,
                      'valu_issue_cycles': issue, 'measured_cycles': measured, 'valu_issue_share': issue / measured,
                      'hbm_frac_of_8TBps': roof['frac']}))


if __name__ == '__main__':
    main()
