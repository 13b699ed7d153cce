#!/usr/bin/env python3
"""Where the mixed-precision form's tail of relative errors comes from: for a synthetic tile, the
relative error of the float32 mixed totals against the float64 arithmetic, against the cancellation
of s A + rho Cp vpd / r_a in the wet-canopy and bare-soil numerators (|sum| / |s A|), per period.
Prints the share of pixels and of the > 1e-5 errors by cancellation class -- the data behind the
decision how (whether) to repair the tail."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mod16_amd import _lib  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    n = 1 << 24
    em = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    ef = RasterEngine(table, dtype='float32', math=_lib.MATH_FAST)
    cls, drv = em.synth(n, seed=16)
    md, mn = em.run(cls, drv)
    fd, fn = ef.run(cls, drv)
    sep = ef.empty(n, 6)
    ef.run(cls, drv, None, None, out_sep=sep)
    d = [x.double() for x in drv]
    lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_ann, tmin, vpd_d, vpd_n, pa, fpar, lai = d
    par = torch.from_numpy(table).cuda()[cls.long()]          # [n, 11]
    gl_sh = par[:, 4]

    def period(t, vpd, rad_net):
        tc = t - 273.15
        esat = 610.8 * torch.exp(17.27 * tc / (tc + 237.3))
        rh = torch.clamp((esat - vpd) / esat, 0, 1)
        fwet = torch.where(rh < 0.7, torch.zeros_like(rh), rh ** 4)
        s = 17.38 * 239.0 * esat / (239.0 + tc) ** 2
        rho = (0.348444 * pa / 100 - rh * 100 * (0.00252 * tc - 0.020582)) / t
        rr = rho * 1013.0 / (4 * 5.67e-8 * t ** 3)
        fw = torch.where(fwet == 0, torch.full_like(fwet, 1e-7), fwet)
        lw = torch.where(lai == 0, torch.full_like(lai, 1e-7), lai)
        g_a = gl_sh * lw * fw + 1.0 / rr
        sA = s * fpar * rad_net
        t2 = rho * 1013.0 * fpar * vpd * g_a
        return (sA + t2).abs() / sA.abs()

    ratio_n = period(t_n, vpd_n, sw_n * (1 - alb) + lw_n)
    ratio_d = period(t_d, vpd_d, sw_d * (1 - alb) + lw_d)
    for name, got, ref, ratio, comp in (('night', mn, fn, ratio_n, sep[3]), ('day', md, fd, ratio_d, sep[0])):
        g, r = got.double(), ref.double()
        ok = torch.isfinite(r) & (r != 0)
        rel = torch.zeros_like(r)
        rel[ok] = ((g[ok] - r[ok]).abs() / r[ok].abs())
        bad = rel > 1e-5
        wet = comp.double() != 0                       # wet-canopy evaporation present
        print('%s: %d of %d values off by > 1e-5 (%.4f %%); of those %d have wet-canopy evaporation'
              % (name, int(bad.sum()), n, 100.0 * float(bad.sum()) / n, int((bad & wet).sum())))
        for k in (4, 16, 64, 256, 1024):
            flag = (ratio < 1.0 / k) & wet
            print('   canopy sum cancels below 1/%-5d: %8d pixels (%.3f %%), hold %6d of the bad ones'
                  % (k, int(flag.sum()), 100.0 * float(flag.sum()) / n, int((bad & flag).sum())))
        small = r.abs() < 1e-3 * float(torch.nan_to_num(r).abs().max())
        print('   bad values below 1e-3 of the largest total: %d' % int((bad & small).sum()))


if __name__ == '__main__':
    main()
