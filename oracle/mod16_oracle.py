"""
TEST INFRASTRUCTURE ONLY -- the parity oracle for the MOD16 forward run.

A CPU (numpy) restatement of the reference's *instance* path
``MOD16.evapotranspiration()`` (reference ``mod16/__init__.py:675-793`` and its
callees). It deliberately keeps the reference's operation order (one numpy
ufunc per arithmetic step, no strength reduction) so that its results are
bit-identical to the reference on the same inputs; every function cites the
reference lines it follows.

Who may use this module: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the reported CPU
baseline, never as the thing shipped. The product (``mod16_amd``) does not
import it and has no CPU fallback.

Pinning (see DESIGN.md "Oracle"): checked against
  * every known-answer test in the reference's ``tests/tests.py`` (restated in
    ``tests/test_oracle_kat.py``),
  * golden vectors made by importing the reference itself in the build
    container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).

Third-party arithmetic: the reference calls ``mod17.linear_constraint``
(``mod16/__init__.py:104,1148-1149``; ``mod17>=0.1.1`` per the reference's
``pyproject.toml:22``, not vendored, not installed here). ``linear_constraint``
below restates its published behaviour (reference ``README.md:351-369``).
"""
import numpy as np

# Reference mod16/__init__.py:106-110
PFT_VALID = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)
STEFAN_BOLTZMANN = 5.67e-8
SPECIFIC_HEAT_CAPACITY_AIR = 1013
MOL_WEIGHT_WET_DRY_RATIO_AIR = 0.622
TINY = 1e-7  # mod16/__init__.py:869,1157

# Reference mod16/__init__.py:152-155 (order matters: it is the column order
# of the BPLUT handed to the HIP library)
PARAM_NAMES = (
    'tmin_close', 'tmin_open', 'vpd_open', 'vpd_close', 'gl_sh', 'gl_wv',
    'g_cuticular', 'csl', 'rbl_min', 'rbl_max', 'beta')

# Driver order of MOD16.evapotranspiration(), mod16/__init__.py:675-682
DRIVER_NAMES = (
    'lw_net_day', 'lw_net_night', 'sw_rad_day', 'sw_rad_night', 'sw_albedo',
    'temp_day', 'temp_night', 'temp_annual', 'tmin', 'vpd_day', 'vpd_night',
    'pressure', 'fpar', 'lai')


def linear_constraint(xmin, xmax, form=None):
    """mod17.linear_constraint restated (reference README.md:351-369; call
    sites mod16/__init__.py:1148-1149). Returns a ramp function of x:
    rising 0..1 over [xmin, xmax), or falling 1..0 when form='reversed'."""
    if form == 'reversed':
        return lambda x: np.where(
            x >= xmax, 0,
            np.where(x < xmin, 1,
                     1 - np.divide(np.subtract(x, xmin), xmax - xmin)))
    return lambda x: np.where(
        x >= xmax, 1,
        np.where(x < xmin, 0, np.divide(np.subtract(x, xmin), xmax - xmin)))


def latent_heat_vaporization(temp_k):
    """mod16/__init__.py:121"""
    return (2.501 - 0.002361 * (temp_k - 273.15)) * 1e6


def svp(temp_k):
    """mod16/__init__.py:1340-1367"""
    temp_c = temp_k - 273.15
    return 1e3 * 0.6108 * np.exp((17.27 * temp_c) / (temp_c + 237.3))


def svp_slope(temp_k, s=None):
    """mod16/__init__.py:1370-1397 (note the constants differ from svp())"""
    if s is None:
        s = svp(temp_k)
    return 17.38 * 239.0 * s / (239.0 + temp_k - 273.15)**2


def psychrometric_constant(pressure, temp_k):
    """mod16/__init__.py:1261-1290"""
    lhv = latent_heat_vaporization(temp_k)
    return (SPECIFIC_HEAT_CAPACITY_AIR * pressure) / \
        (lhv * MOL_WEIGHT_WET_DRY_RATIO_AIR)


def air_density(temp_k, pressure, rhumidity):
    """mod16/__init__.py:384-412"""
    return np.divide(
        0.348444 * (pressure / 100) - (rhumidity * 100) *
        (0.00252 * (temp_k - 273.15) - 0.020582),
        temp_k)


def rhumidity(temp_k, vpd):
    """mod16/__init__.py:646-673"""
    esat = svp(temp_k)
    avp = esat - vpd
    rh = avp / esat
    return np.where(avp < 0, 0, np.where(rh > 1, 1, rh))


def wet_fraction(rh):
    """mod16/__init__.py:764"""
    return np.where(rh < 0.7, 0, np.power(rh, 4))


def r_correction(pressure, temp_k):
    """mod16/__init__.py:771"""
    return (101300 / pressure) * (temp_k / 293.15)**1.75


def radiative_resistance(rho, temp_k):
    """mod16/__init__.py:947-948 (identical at :519-520, :1231-1232)"""
    return (rho * SPECIFIC_HEAT_CAPACITY_AIR) / (
        4 * STEFAN_BOLTZMANN * temp_k**3)


def soil_heat_flux(p, rad_net_day, rad_net_night, temp_day, temp_night,
                   temp_annual):
    """mod16/__init__.py:1055-1119"""
    condition = np.logical_and(
        np.logical_and(
            temp_annual < (273.15 + 25),
            temp_annual >= (273.15 + p['tmin_close'])),
        (temp_day - temp_night) >= 5)
    out = []
    for rad_i, temp_i in ((rad_net_day, temp_day), (rad_net_night, temp_night)):
        g = np.where(condition, (4.73 * (temp_i - 273.15)) - 20.87, 0)
        g = np.where(np.abs(g) > (0.39 * np.abs(rad_i)), 0.39 * rad_i, g)
        out.append(g)
    return out


def radiation_soil(p, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                   sw_albedo, temp_day, temp_night, temp_annual, fpar):
    """mod16/__init__.py:963-1053"""
    rad_net_day = sw_rad_day * (1 - sw_albedo) + lw_net_day
    rad_net_night = lw_net_night
    g_day, g_night = soil_heat_flux(
        p, rad_net_day, rad_net_night, temp_day, temp_night, temp_annual)
    g_day = np.where(
        np.logical_and(rad_net_day - g_day < 0, rad_net_day > 0),
        rad_net_day, g_day)
    g_night = np.where(
        np.logical_and(
            rad_net_day > 0,
            (rad_net_night - g_night) < (-0.5 * rad_net_day)),
        rad_net_night + (0.5 * rad_net_day), g_night)
    rad_soil_day = (1 - fpar) * (rad_net_day - g_day)
    rad_soil_night = (1 - fpar) * (rad_net_night - g_night)
    return (rad_soil_day, rad_soil_night)


def evaporation_wet_canopy(p, pressure, temp_k, vpd, lai, fpar, rad_canopy,
                           lhv=None, rh=None, f_wet=None, tiny=TINY):
    """mod16/__init__.py:866-961"""
    if lhv is None:
        lhv = latent_heat_vaporization(temp_k)
    if rh is None:
        rh = rhumidity(temp_k, vpd)
    if f_wet is None:
        f_wet = wet_fraction(rh)
    f_wet = np.where(f_wet == 0, f_wet + tiny, f_wet)
    lai = np.where(lai == 0, lai + tiny, lai)
    s = svp_slope(temp_k)
    rho = air_density(temp_k, pressure, rh)
    with np.errstate(all='ignore'):
        r_h = 1 / (p['gl_sh'] * lai * f_wet)
        r_e = 1 / (p['gl_wv'] * lai * f_wet)
        r_r = radiative_resistance(rho, temp_k)
        r_a_wet = np.divide(r_h * r_r, r_h + r_r)
        numer = f_wet * ((s * rad_canopy) + (
            rho * SPECIFIC_HEAT_CAPACITY_AIR * fpar * vpd * 1 / r_a_wet))
        denom = s + ((pressure * SPECIFIC_HEAT_CAPACITY_AIR * r_e) *
                     1 / (lhv * MOL_WEIGHT_WET_DRY_RATIO_AIR * r_a_wet))
        evap = np.where(numer < 0, 0, (numer / denom) / lhv)
    return np.where(np.logical_or(f_wet <= tiny, lai <= tiny), 0, evap)


def potential_soil_evaporation(p, pressure, temp_k, vpd, fpar, rad_soil,
                               r_corr=None, lhv=None, rh=None, f_wet=None):
    """mod16/__init__.py:449-544"""
    if lhv is None:
        lhv = latent_heat_vaporization(temp_k)
    if rh is None:
        rh = rhumidity(temp_k, vpd)
    if f_wet is None:
        f_wet = wet_fraction(rh)
    if r_corr is None:
        r_corr = r_correction(pressure, temp_k)
    s = svp_slope(temp_k)
    rho = air_density(temp_k, pressure, rh)
    gamma = psychrometric_constant(pressure, temp_k)
    r_r = radiative_resistance(rho, temp_k)
    r_tot = np.where(
        vpd <= p['vpd_open'], p['rbl_min'],
        np.where(
            vpd >= p['vpd_close'], p['rbl_max'],
            p['rbl_max'] - (
                (p['rbl_max'] - p['rbl_min']) * (p['vpd_close'] - vpd))
            / (p['vpd_close'] - p['vpd_open'])))
    r_tot = r_tot / r_corr
    r_as = (r_tot * r_r) / (r_tot + r_r)
    numer = (s * rad_soil) + \
        (rho * SPECIFIC_HEAT_CAPACITY_AIR * (1 - fpar) * (vpd / r_as))
    denom = (s + gamma * (r_tot / r_as))
    evap_sat = (numer * f_wet) / denom
    evap_unsat = (numer * (1 - f_wet)) / denom
    return (evap_sat, evap_unsat)


def evaporation_soil(p, pressure, temp_k, vpd, fpar, rad_soil, r_corr=None,
                     lhv=None, rh=None, f_wet=None):
    """mod16/__init__.py:795-864"""
    if lhv is None:
        lhv = latent_heat_vaporization(temp_k)
    if rh is None:
        rh = rhumidity(temp_k, vpd)
    evap_sat, evap_unsat = potential_soil_evaporation(
        p, pressure, temp_k, vpd, fpar, rad_soil, r_corr, lhv, rh, f_wet)
    e = np.where(evap_sat < 0, 0, evap_sat)
    e = e + np.where(
        evap_unsat < 0, 0, evap_unsat * np.power(rh, vpd / p['beta']))
    return e / lhv


def surface_conductance(p, tmin, vpd_day):
    """mod16/__init__.py:1121-1150"""
    m_tmin = linear_constraint(p['tmin_close'], p['tmin_open'])
    m_vpd = linear_constraint(p['vpd_open'], p['vpd_close'], 'reversed')
    return (p['csl'] * m_tmin(tmin - 273.15) * m_vpd(vpd_day))


def transpiration(p, pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin,
                  r_corr=None, lhv=None, rh=None, f_wet=None, daytime=True,
                  tiny=TINY):
    """mod16/__init__.py:1152-1258"""
    if lhv is None:
        lhv = latent_heat_vaporization(temp_k)
    if rh is None:
        rh = rhumidity(temp_k, vpd)
    if f_wet is None:
        f_wet = wet_fraction(rh)
    if r_corr is None:
        r_corr = r_correction(pressure, temp_k)
    s = svp_slope(temp_k)
    rho = air_density(temp_k, pressure, rh)
    gamma = psychrometric_constant(pressure, temp_k)
    r_r = radiative_resistance(rho, temp_k)
    g_surf = 0
    if daytime:
        g_surf = surface_conductance(p, tmin, vpd) / r_corr
    g_cuticular = p['g_cuticular'] / r_corr
    gl_sh = p['gl_sh'] * lai * (1 - f_wet)
    g = ((gl_sh * (g_surf + g_cuticular)) / (gl_sh + g_surf + g_cuticular))
    g_canopy = np.where(np.logical_and(lai > 0, (1 - f_wet) > 0), g, tiny)
    r_a_dry = (1 / p['gl_sh'] * r_r) / (1 / p['gl_sh'] + r_r)
    rad_canopy = np.where(rad_canopy < 0, 0, rad_canopy)
    t = (1 - f_wet) * ((s * rad_canopy) + (
        rho * SPECIFIC_HEAT_CAPACITY_AIR * fpar * (vpd / r_a_dry)))
    t = t / (s + gamma * (1 + (1 / g_canopy) / r_a_dry))
    return np.where(g_canopy <= tiny, 0, t / lhv)


def evapotranspiration(p, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                       sw_albedo, temp_day, temp_night, temp_annual, tmin,
                       vpd_day, vpd_night, pressure, fpar, lai, f_wet=None,
                       separate=False):
    """mod16/__init__.py:675-793. ``p`` is a dict of the 11 parameters
    (scalars or arrays broadcastable against the drivers). As in the
    reference, the caller's ``f_wet`` is ignored (:764)."""
    with np.errstate(all='ignore'):
        rad_soil = radiation_soil(
            p, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
            temp_day, temp_night, temp_annual, fpar)
        out = []
        for i, (temp_k, vpd, sw_rad, lw_net) in enumerate((
                (temp_day, vpd_day, sw_rad_day, lw_net_day),
                (temp_night, vpd_night, sw_rad_night, lw_net_night))):
            rad_net = sw_rad * (1 - sw_albedo) + lw_net
            rad_canopy = fpar * rad_net
            rh = rhumidity(temp_k, vpd)
            fw = wet_fraction(rh)
            lhv = latent_heat_vaporization(temp_k)
            r_corr = r_correction(pressure, temp_k)
            e_canopy = evaporation_wet_canopy(
                p, pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv, rh, fw)
            e_soil = evaporation_soil(
                p, pressure, temp_k, vpd, fpar, rad_soil[i], r_corr, lhv, rh,
                fw)
            trans = transpiration(
                p, pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin, r_corr,
                lhv, rh, fw, daytime=(i == 0))
            if separate:
                out.append((e_canopy, e_soil, trans))
            else:
                out.append(e_canopy + e_soil + trans)
    return tuple(out)


def potential_transpiration(lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd,
                            fpar, rh=None, f_wet=None, alpha=1.26):
    """MOD16.potential_transpiration, mod16/__init__.py:546-602 [W m-2]"""
    if rh is None:
        rh = rhumidity(temp_k, vpd)
    if f_wet is None:
        f_wet = wet_fraction(rh)
    rad_net = sw_rad * (1 - sw_albedo) + lw_net
    rad_canopy = fpar * rad_net
    s = svp_slope(temp_k)
    gamma = psychrometric_constant(pressure, temp_k)
    return (alpha * (s * rad_canopy) * (1 - f_wet)) / (s + gamma)


def radiation_net(sw_rad, sw_albedo, temp_k):
    """radiation_net (DEPRECATED in the reference), mod16/__init__.py:1293-1337"""
    emis_surface = 0.97
    emis_atmos = 1 - 0.26 * np.exp(-7.77e-4 * np.power(temp_k - 273.15, 2))
    return sw_rad * (1 - sw_albedo) + \
        STEFAN_BOLTZMANN * (emis_atmos - emis_surface) * np.power(temp_k, 4)


def potential_et(p, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                 sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                 vpd_night, pressure, fpar, lai):
    """Potential ET (day, night) [kg m-2 s-1] as the reference's README defines
    it (README.md:404-424), composed from the reference's own component
    methods: wet-canopy evaporation (:866) + potential soil evaporation of the
    saturated and unsaturated fractions (:449, clamped at 0 as :858-861 does,
    no soil-moisture constraint) + potential transpiration (:546)."""
    with np.errstate(all='ignore'):
        rad_soil = radiation_soil(
            p, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
            temp_day, temp_night, temp_annual, fpar)
        out = []
        for i, (temp_k, vpd, sw_rad, lw_net) in enumerate((
                (temp_day, vpd_day, sw_rad_day, lw_net_day),
                (temp_night, vpd_night, sw_rad_night, lw_net_night))):
            rad_canopy = fpar * (sw_rad * (1 - sw_albedo) + lw_net)
            rh = rhumidity(temp_k, vpd)
            fw = wet_fraction(rh)
            lhv = latent_heat_vaporization(temp_k)
            r_corr = r_correction(pressure, temp_k)
            canopy = evaporation_wet_canopy(
                p, pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv, rh, fw)
            sat, unsat = potential_soil_evaporation(
                p, pressure, temp_k, vpd, fpar, rad_soil[i], r_corr, lhv, rh, fw)
            e = np.where(sat < 0, 0, sat)
            e = e + np.where(unsat < 0, 0, unsat)
            ptr = potential_transpiration(
                lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd, fpar, rh, fw)
            out.append((canopy + e / lhv) + ptr / lhv)
    return tuple(out)


def vpd_from_humidity(qv10m, pressure, tmean):
    """MOD16.vpd, mod16/__init__.py:604-644 (its own SVP constants)"""
    temp_c = tmean - 273.15
    avp = (qv10m * pressure) / (0.622 + (0.379 * qv10m))
    sv = 610.7 * np.exp((17.38 * temp_c) / (239 + temp_c))
    return sv - avp


def air_pressure(elevation_m):
    """MOD16.air_pressure, mod16/__init__.py:414-447"""
    rate = 9.80665 / (0.0065 * (8.3143 / 28.9644e-3))
    temp_ratio = 1 - ((0.0065 * elevation_m) / 288.15)
    return 101325.0 * np.power(temp_ratio, rate)


def evapotranspiration_raw(bplut, cls, raw, fpar_pct, lai_x10, day_hours=None):
    """Forward run on raw drivers: the reference's pre-processing
    (mod16/calibration.py:380-423) followed by the instance path; with
    ``day_hours`` also the 8-day total of tests/verification/verify2.py:113-115.
    ``raw`` = the 14 arrays in mod16_raw_driver order."""
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, qv_d, qv_n, ps_d, ps_n, elev) = raw
    with np.errstate(all='ignore'):
        vpd_d = vpd_from_humidity(qv_d, ps_d, t_d)
        vpd_n = vpd_from_humidity(qv_n, ps_n, t_n)
        vpd_n = np.where(vpd_n < 0, 0, vpd_n)
        pa = air_pressure(elev)
        fpar = np.where(fpar_pct >= 249, np.nan, np.asarray(fpar_pct, t_d.dtype)) / 100
        lai = np.where(lai_x10 >= 249, np.nan, np.asarray(lai_x10, t_d.dtype)) / 10
        day, night = evapotranspiration_raster(
            bplut, cls, lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa,
            fpar, lai)
        if day_hours is None:
            return day, night
        total = ((day * day_hours * 8 * 60 * 60) + (night * (24 - day_hours) * 8 * 60 * 60))
    return day, night, total


def gather_params(bplut, cls):
    """The multi-class idiom of the reference's forward-run notebook (cell 32):
    ``params_dict[key][pft_map]`` per parameter. ``bplut`` maps the 11
    parameter names to arrays of length 13 (as ``restore_bplut`` returns,
    reference mod16/utils.py:81-117). numpy raises IndexError for a class
    >= 13, as the reference idiom would."""
    cls = np.asarray(cls)
    return {k: np.asarray(bplut[k])[cls] for k in PARAM_NAMES}


def evapotranspiration_raster(bplut, cls, *drivers, separate=False):
    """ET over a multi-class raster: per-pixel parameter gather followed by
    the instance path, i.e. exactly
    ``MOD16({k: bplut[k][cls] ...}).evapotranspiration(*drivers)``."""
    return evapotranspiration(
        gather_params(bplut, cls), *drivers, separate=separate)


def bplut_table(bplut):
    """dict of 11 arrays(13) -> C-contiguous float64 [13][11] table in
    PARAM_NAMES column order (the layout ``mod16_set_bplut_f64`` takes)."""
    return np.ascontiguousarray(
        np.stack([np.asarray(bplut[k], np.float64) for k in PARAM_NAMES], 1))


# ---------------------------------------------------------------------------
# The vectorised calibration path. It is a
# different algorithm from the instance path above (W m-2, other clamps, a
# whole-array branch on g_surf), restated here in its own operation order.

def et_static_daynight(params, lw_net_day, lw_net_night, sw_rad_day,
                       sw_rad_night, sw_albedo, temp_day, temp_night,
                       temp_annual, tmin, vpd_day, vpd_night, pressure, fpar,
                       lai, tiny=TINY, r_corr_list=None):
    """MOD16._evapotranspiration, mod16/__init__.py:195-382: [day, night]
    latent heat flux [W m-2]. ``params`` is the 11-vector in
    MOD16.required_parameters order (scalars or (1 x N) arrays)."""
    with np.errstate(all='ignore'):
        rad_net_day = sw_rad_day * (1 - sw_albedo) + lw_net_day          # :225
        rad_net_night = lw_net_night
        condition = np.logical_and(                                       # :230-234
            np.logical_and(
                temp_annual < (273.15 + 25),
                temp_annual > (273.15 + params[1])),
            (temp_day - temp_night) >= 5)
        g_soil = []
        for rad_i, temp_i in ((rad_net_day, temp_day), (rad_net_night, temp_night)):
            g = np.where(condition, (4.73 * (temp_i - 273.15)) - 20.87, 0)
            g = np.where(np.abs(g) > (0.39 * np.abs(rad_i)), 0.39 * rad_i, g)
            g_soil.append(g)
        g_soil_day, g_soil_night = g_soil
        g_soil_day = np.where(
            np.logical_and(rad_net_day - g_soil_day < 0, rad_net_day > 0),
            rad_net_day, g_soil_day)
        g_soil_night = np.where(
            np.logical_and(
                rad_net_day > 0,
                (rad_net_night - g_soil_night) < (-0.5 * rad_net_day)),
            rad_net_night + (0.5 * rad_net_day), g_soil_night)
        rad_soil_day = (1 - fpar) * (rad_net_day - g_soil_day)
        rad_soil_night = (1 - fpar) * (rad_net_night - g_soil_night)
        out = []
        for i, (temp_k, vpd, sw_rad, lw_net, rad_soil) in enumerate((
                (temp_day, vpd_day, sw_rad_day, lw_net_day, rad_soil_day),
                (temp_night, vpd_night, sw_rad_night, lw_net_night, rad_soil_night))):
            daytime = (i == 0)
            rad_net = sw_rad * (1 - sw_albedo) + lw_net
            rad_canopy = fpar * rad_net
            _svp = svp(temp_k)
            rh = (_svp - vpd) / _svp                                      # :280-281
            rh = np.where(rh < 0, 0, rh)
            f_wet = np.where(rh < 0.7, 0, np.power(rh, 4))
            s = svp_slope(temp_k, _svp)
            lhv = latent_heat_vaporization(temp_k)
            gamma = psychrometric_constant(pressure, temp_k)
            if r_corr_list is None:
                r_corr = (101300 / pressure) * (temp_k / 293.15)**1.75
            else:
                r_corr = r_corr_list[i]
            rho = air_density(temp_k, pressure, rh)
            r_r = (rho * SPECIFIC_HEAT_CAPACITY_AIR) / (
                4 * STEFAN_BOLTZMANN * temp_k**3)
            r_h = 1 / (params[4] * lai * f_wet)                           # :305-311
            r_e = 1 / (params[5] * lai * f_wet)
            r_a_wet = np.divide(r_h * r_r, r_h + r_r)
            e = np.divide(
                f_wet * ((s * rad_canopy) + (
                    rho * SPECIFIC_HEAT_CAPACITY_AIR * fpar * vpd * 1 / r_a_wet)),
                s + ((pressure * SPECIFIC_HEAT_CAPACITY_AIR * r_e) *
                     1 / (lhv * MOL_WEIGHT_WET_DRY_RATIO_AIR * r_a_wet)))
            e_canopy = np.where(lai * f_wet <= tiny, 0, e)                # :320
            g_surf = 0
            if daytime:
                m_tmin = linear_constraint(params[0], params[1])
                m_vpd = linear_constraint(params[2], params[3], 'reversed')
                g_surf = (params[7] * m_tmin(tmin - 273.15) * m_vpd(vpd))
            g_surf = g_surf / r_corr                                      # :328
            g_cuticular = params[6] / r_corr
            gl_sh = params[4] * lai * (1 - f_wet)
            g = ((gl_sh * (g_surf + g_cuticular)) / (
                gl_sh + g_surf + g_cuticular))
            g_canopy = np.where(
                np.logical_and(lai > 0, (1 - f_wet) > 0), g, tiny)
            r_a_dry = (1 / params[4] * r_r) / (1 / params[4] + r_r)
            if np.any(g_surf > 0):                                        # :343-348
                t = (1 - f_wet) * ((s * rad_canopy) + (
                    rho * SPECIFIC_HEAT_CAPACITY_AIR * fpar * (vpd / r_a_dry)))
                t = t / (s + gamma * (1 + (1 / g_canopy) / r_a_dry))
            else:
                t = 0
            r_tot = np.where(
                vpd <= params[2], params[8],
                np.where(vpd >= params[3], params[9],
                         params[9] - ((params[9] - params[8]) * (params[3] - vpd))
                         / (params[3] - params[2])))
            r_tot = r_tot / r_corr
            r_as = (r_tot * r_r) / (r_tot + r_r)
            numer = (s * rad_soil) + (
                rho * SPECIFIC_HEAT_CAPACITY_AIR * (1 - fpar) * (vpd / r_as))
            denom = (s + gamma * (r_tot / r_as))
            evap_sat = (numer * f_wet) / denom
            evap_unsat = (numer * (1 - f_wet)) / denom
            e_soil = evap_sat + evap_unsat * rh**(vpd / params[10])       # :376
            out.append((t + e_canopy + e_soil))                           # :380
    return out


def et_static(params, *drivers, r_corr_list=None):
    """MOD16._et, mod16/__init__.py:162-193: day + night [W m-2]."""
    day, night = et_static_daynight(params, *drivers, r_corr_list=r_corr_list)
    return np.add(day, night)
