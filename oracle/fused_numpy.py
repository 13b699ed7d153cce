"""
TEST / BENCH INFRASTRUCTURE ONLY -- the second CPU baseline of SURVEY.md section 8d:
the MOD16 forward run as FUSED, strength-reduced numpy.

``oracle/mod16_oracle.py`` keeps the reference's ~500-pass operation sequence (it is the
parity oracle). This module computes the same totals the way the GPU kernel's FAST form
does (``mod16_amd/csrc/mod16_physics.hpp``, ``period_fast``): per period the saturation
vapour pressure, its slope, the latent heat, rho Cp, the radiative conductance and
1 / r_corr once, shared by the three components; parallel resistances as conductances (one
division per component); x^1.75 through square roots; rh^4 by squaring; in-place ufuncs
(``out=``) wherever a temporary can be reused. Every ``np.where`` predicate of the reference
is kept, so NaN and exact-zero masks are the oracle's. It is NOT a parity reference -- it is
checked against the oracle (``tests/test_oracle_golden.py``: 1e-9, identical masks) and
timed by ``bench.py`` as ``cpu_baseline.fused_numpy``: the best a numpy caller could do
without leaving numpy.
"""
import numpy as np

from . import mod16_oracle as oracle


def _period(p, sh, t, vpd, rad_net, rad_soil, daytime):
    tiny = oracle.TINY
    tc = t - 273.15
    esat = 610.8 * np.exp((17.27 * tc) / (tc + 237.3))                # :1340-1367
    avp = esat - vpd
    rh = avp / esat
    rh = np.where(avp < 0, 0, np.where(rh > 1, 1, rh))                # :670-673
    dry = rh < 0.7
    rh2 = rh * rh
    fwet = np.where(dry, 0, rh2 * rh2)                                # :764
    omw = 1 - fwet
    ta = (239.0 + t) - 273.15
    s = (17.38 * 239.0 * esat) / (ta * ta)                            # :1395-1397
    lhv = (2.501 - 0.002361 * tc) * 1e6                               # :121
    slhv = s * lhv
    x = t * (1.0 / 293.15)
    sq = np.sqrt(x)
    inv_rcorr = sh['p_rel'] / (x * sq * np.sqrt(sq))                  # 1 / ((101300/P) x^1.75), :771
    nn = sh['p_mbar_k'] - (rh * 100) * (0.00252 * tc - 0.020582)      # rho T, :408-412
    u = 1.0 / (nn * t)
    rho_cp = 1013.0 * ((nn * nn) * u)
    t2 = t * t
    g_rr = (4 * 5.67e-8 / 1013.0) * ((t2 * t2) * t) * u               # 1 / r_r, :947
    rcfv = rho_cp * vpd
    fpar = sh['fpar']
    radc = fpar * rad_net
    # wet canopy, :866-961, in conductances
    fw = np.where(dry, tiny, fwet)
    g_e = sh['glwv_l'] * fw
    g_a = sh['glsh_l'] * fw + g_rr
    numer = fw * ((rcfv * fpar) * g_a + s * radc)
    evap = (numer * g_e) / (slhv * g_e + sh['k_p'] * g_a)
    canopy = np.where((numer < 0) | dry | sh['lai_tiny'], 0, evap)
    # bare soil, :449-544, :795-864
    r0 = np.where(vpd <= p['vpd_open'], p['rbl_min'],
                  np.where(vpd >= p['vpd_close'], p['rbl_max'],
                           p['rbl_max'] - (p['vpd_close'] - vpd) * sh['rbl_slope']))
    r_tot = r0 * inv_rcorr
    w = r_tot * g_rr + 1
    q = ((s * rad_soil) * r_tot + (rcfv * sh['omf']) * w) / (r_tot * (sh['k_p'] * w + slhv))
    pw = np.power(rh, vpd / p['beta'])
    soil = np.where(q < 0, 0, q * (omw * pw + fwet))
    # transpiration, :1152-1258
    if daytime:
        m_vpd = np.where(vpd >= p['vpd_close'], 0,
                         np.where(vpd < p['vpd_open'], 1, 1 - (vpd - p['vpd_open']) * sh['inv_dvpd']))
        g_s = ((p['csl'] * sh['m_tmin']) * m_vpd) * inv_rcorr
    else:
        g_s = 0.0
    gsc = g_s + p['g_cuticular'] * inv_rcorr
    g_bl = sh['glsh_lai'] * omw
    p1 = g_bl * gsc
    s1 = g_bl + gsc
    shut = ~(sh['lai_pos'] & (omw > 0)) | (p1 <= tiny * s1)
    g_d = p['gl_sh'] + g_rr
    rad_c = np.where(radc < 0, 0, radc)
    tr = ((omw * ((rcfv * fpar) * g_d + s * rad_c)) * p1) / (slhv * p1 + sh['k_p'] * (g_d * s1 + p1))
    trans = np.where(shut, 0, tr)
    return (canopy + soil) + trans


def evapotranspiration_raster(bplut, cls, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                              sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                              vpd_night, pressure, fpar, lai):
    """(day, night) totals [kg m-2 s-1] of a multi-class raster; ``bplut``: dict of 11
    arrays of 13, as for ``mod16_oracle.evapotranspiration_raster``."""
    with np.errstate(all='ignore'):
        p = oracle.gather_params(bplut, cls)
        oma = 1 - sw_albedo
        omf = 1 - fpar
        a_d = sw_rad_day * oma + lw_net_day
        a_n = lw_net_night
        cond = (temp_annual < 273.15 + 25.0) & (temp_annual >= (273.15 + p['tmin_close'])) & \
            ((temp_day - temp_night) >= 5)
        g_d = np.where(cond, (4.73 * (temp_day - 273.15)) - 20.87, 0)
        g_d = np.where(np.abs(g_d) > (0.39 * np.abs(a_d)), 0.39 * a_d, g_d)
        g_n = np.where(cond, (4.73 * (temp_night - 273.15)) - 20.87, 0)
        g_n = np.where(np.abs(g_n) > (0.39 * np.abs(a_n)), 0.39 * a_n, g_n)
        g_d = np.where((a_d - g_d < 0) & (a_d > 0), a_d, g_d)
        g_n = np.where((a_d > 0) & ((a_n - g_n) < (-0.5 * a_d)), a_n + (0.5 * a_d), g_n)
        rs_d = omf * (a_d - g_d)
        rs_n = omf * (a_n - g_n)
        l_wet = np.where(lai == 0, oracle.TINY, lai)
        tm = tmin - 273.15
        dv = p['vpd_close'] - p['vpd_open']
        sh = {
            'fpar': fpar, 'omf': omf, 'p_rel': pressure * (1.0 / 101300.0),
            'k_p': pressure * (1013.0 / 0.622), 'p_mbar_k': pressure * (0.348444 / 100.0),
            'lai_tiny': l_wet <= oracle.TINY, 'lai_pos': lai > 0,
            'glsh_l': p['gl_sh'] * l_wet, 'glwv_l': p['gl_wv'] * l_wet, 'glsh_lai': p['gl_sh'] * lai,
            'm_tmin': np.where(tm >= p['tmin_open'], 1,
                               np.where(tm < p['tmin_close'], 0,
                                        (tm - p['tmin_close']) / (p['tmin_open'] - p['tmin_close']))),
            'inv_dvpd': 1.0 / dv, 'rbl_slope': (p['rbl_max'] - p['rbl_min']) / dv,
        }
        day = _period(p, sh, temp_day, vpd_day, a_d, rs_d, True)
        rn_n = sw_rad_night * oma + lw_net_night
        night = _period(p, sh, temp_night, vpd_night, rn_n, rs_n, False)
    return day, night
