#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by running the REFERENCE
(arthur-e/MOD16 v1.2.0, mounted read-only at /root/reference) in the build
container. The reference never travels to the GPU box; only the .npz files
written here (inputs + the reference's outputs) do.

The reference needs ``mod17.linear_constraint`` (``mod16/__init__.py:104``);
``mod17`` is not installed and there is no network. A scratch module named
``mod17`` is therefore created in a temporary directory OUTSIDE the repo; it
re-exports this repo's restatement ``oracle.mod16_oracle.linear_constraint``
(reference README.md:351-369). Everything else that runs is the reference's
own code.

Usage (build container only):  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import mod16_oracle as oracle  # noqa: E402
from oracle import synth  # noqa: E402


def import_reference():
    scratch = tempfile.mkdtemp(prefix='mod17_standin_')
    os.makedirs(os.path.join(scratch, 'mod17'))
    with open(os.path.join(scratch, 'mod17', '__init__.py'), 'w') as f:
        f.write(
            'import sys\n'
            'sys.path.insert(0, %r)\n'
            'from oracle.mod16_oracle import linear_constraint\n' % ROOT)
    sys.path.insert(0, scratch)
    sys.path.insert(0, REFERENCE)
    import mod16
    import mod16.utils
    import mod16.models
    assert mod16.__file__.startswith(REFERENCE), mod16.__file__
    return mod16


def pack_sep(result):
    (c_d, s_d, t_d), (c_n, s_n, t_n) = result
    return dict(canopy_day=c_d, soil_day=s_d, trans_day=t_d,
                canopy_night=c_n, soil_night=s_n, trans_night=t_n)


def make_f9(ref):
    """F9 (round 2): vectors the first set did not hold -- written by
    ``make_golden.py --f9`` alone so that the files of F1-F8 stay byte for byte
    what they were.

    * ``MOD16.potential_transpiration`` (:546-602, default and another alpha,
      with and without the optional rhumidity / f_wet) and the deprecated
      ``radiation_net`` (:1293-1337);
    * ``evaporation_wet_canopy`` / ``transpiration`` / ``_evapotranspiration``
      with ``tiny`` other than 1e-7 (:869, :1157, :199), on inputs with lai = 0
      and wet fractions that are exactly 0 or below the new ``tiny``;
    * the (N,)-against-(T, N) broadcast of the forward-run notebook (cell 17):
      per-site ``pressure`` / ``temp_annual`` rows, a scalar ``sw_rad_night``
      and a (T, 1) column against (T, N) drivers, through
      ``MOD16.evapotranspiration`` with per-site (N,) parameter arrays.
    """
    MOD16 = ref.MOD16
    names = list(MOD16.required_parameters)
    data_dir = os.path.join(REFERENCE, 'mod16', 'data')
    c51 = 'MOD16_BPLUT_C5.1_05deg_MCD43B_Albedo_MERRA_GMAO.csv'
    bplut = ref.utils.restore_bplut(os.path.join(data_dir, c51))
    bplut['beta'] = np.where(np.isnan(bplut['tmin_close']), np.nan, 250.0)
    p2 = {k: bplut[k][7] for k in names}
    m2 = MOD16(p2)
    _, drv = synth.drivers((256,), seed=9, special=False)
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin_, vpd_d, vpd_n, pa,
     fpar_, lai_) = drv
    lai_ = lai_.copy()
    lai_[:8] = 0.0
    lai_[8:16] = 5e-4            # below tiny = 1e-3, above the default
    vpd_d = vpd_d.copy()
    vpd_d[16:24] = 0.0           # rh = 1 -> f_wet = 1 -> (1 - f_wet) = 0
    f9 = dict(params=np.array([p2[k] for k in names], float),
              drivers=np.stack([lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin_,
                                vpd_d, vpd_n, pa, fpar_, lai_]))
    rh = MOD16.rhumidity(t_d, vpd_d)
    fw = np.where(rh < 0.7, 0.0, rh ** 4)
    f9['rhumidity'], f9['f_wet'] = rh, fw
    f9['potential_transpiration'] = MOD16.potential_transpiration(
        lw_d, sw_d, alb, pa, t_d, vpd_d, fpar_)
    f9['potential_transpiration_alpha1'] = MOD16.potential_transpiration(
        lw_d, sw_d, alb, pa, t_d, vpd_d, fpar_, alpha=1.0)
    f9['potential_transpiration_given'] = MOD16.potential_transpiration(
        lw_d, sw_d, alb, pa, t_d, vpd_d, fpar_, rhumidity=rh, f_wet=fw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        f9['radiation_net'] = ref.radiation_net(sw_d, alb, t_d)
        rad_c = fpar_ * (sw_d * (1 - alb) + lw_d)
        f9['rad_canopy'] = rad_c
        for tag, tiny in (('1e-3', 1e-3), ('1e-12', 1e-12)):
            f9['wet_canopy_tiny_' + tag] = m2.evaporation_wet_canopy(
                pa, t_d, vpd_d, lai_, fpar_, rad_c, tiny=tiny)
            f9['transpiration_day_tiny_' + tag] = m2.transpiration(
                pa, t_d, vpd_d, lai_, fpar_, rad_c, tmin_, daytime=True, tiny=tiny)
            f9['transpiration_night_tiny_' + tag] = m2.transpiration(
                pa, t_n, vpd_n, lai_, fpar_, fpar_ * lw_n, tmin_, daytime=False, tiny=tiny)
        # the calibration path with another tiny
        T_, N_ = 6, 24
        cls7, drv7 = synth.drivers((T_, N_), seed=19, special=False)
        site_cls = cls7[0]
        pars7 = [bplut[k][site_cls].reshape(1, N_) for k in names]
        drv7[13][2, :6] = 0.0
        drv7[13][3, :6] = 2e-3
        f9['static_params'] = np.concatenate(pars7, 0)
        f9['static_drivers'] = np.stack(drv7)
        d, n_ = MOD16._evapotranspiration(pars7, *drv7, tiny=1e-2)
        f9['static_day_tiny_1e-2'], f9['static_night_tiny_1e-2'] = d, n_
        f9['static_et_ignores_tiny'] = MOD16._et(pars7, *drv7, tiny=1e-2)
        # notebook cell 17: (N,) rows, a scalar and a (T, 1) column against (T, N)
        T_, N_ = 9, 31
        cls17, drv17 = synth.drivers((T_, N_), seed=17, special=True)
        site_cls = cls17[0]
        site_par = {k: bplut[k][site_cls] for k in names}            # (N,)
        mixed = list(drv17)
        mixed[3] = 0                                                  # sw_rad_night scalar
        mixed[7] = drv17[7][0].copy()                                 # temp_annual (N,)
        mixed[11] = drv17[11][0].copy()                               # pressure (N,)
        mixed[4] = drv17[4][:, :1].copy()                             # albedo (T, 1)
        day, night = MOD16(site_par).evapotranspiration(*mixed)
        sep = pack_sep(MOD16(site_par).evapotranspiration(*mixed, separate=True))
    f9['bcast_site_cls'] = site_cls
    f9['bcast_site_params'] = np.stack([site_par[k] for k in names])
    f9['bcast_dense'] = np.stack([drv17[k] for k in (0, 1, 2, 5, 6, 8, 9, 10, 12, 13)])
    f9['bcast_temp_annual'], f9['bcast_pressure'] = mixed[7], mixed[11]
    f9['bcast_albedo'] = mixed[4]
    f9['bcast_day'], f9['bcast_night'] = day, night
    for k, v in sep.items():
        f9['bcast_' + k] = v
    np.savez_compressed(os.path.join(HERE, 'f9_round2.npz'), **f9)
    print('f9_round2.npz', os.path.getsize(os.path.join(HERE, 'f9_round2.npz')), 'bytes')


def main():
    ref = import_reference()
    if '--f9' in sys.argv:
        return make_f9(ref)
    MOD16 = ref.MOD16
    names = list(MOD16.required_parameters)
    assert tuple(names) == oracle.PARAM_NAMES
    out = {}

    # ---- BPLUT tables as the reference parses them (mod16/utils.py:81-117)
    data_dir = os.path.join(REFERENCE, 'mod16', 'data')
    bplut_files = sorted(f for f in os.listdir(data_dir) if f.endswith('.csv'))
    bpluts = {}
    for fn in bplut_files:
        d = ref.utils.restore_bplut(os.path.join(data_dir, fn))
        bpluts[fn] = np.stack([d[k] for k in names], 1)   # [13][11]
    np.savez(os.path.join(HERE, 'bplut_tables.npz'), **bpluts)
    # MOD16Collection61 parameters per PFT (mod16/models.py:36-51)
    c61 = np.full((13, 11), np.nan)
    for pft in ref.PFT_VALID:
        m = ref.models.MOD16Collection61(pft)
        c61[pft] = [getattr(m, k) for k in names]
    np.savez(os.path.join(HERE, 'collection61_params.npz'), table=c61)

    c51 = 'MOD16_BPLUT_C5.1_05deg_MCD43B_Albedo_MERRA_GMAO.csv'
    bplut = ref.utils.restore_bplut(os.path.join(data_dir, c51))
    bplut['beta'] = np.where(np.isnan(bplut['tmin_close']), np.nan, 250.0)
    table = np.stack([bplut[k] for k in names], 1)

    # ---- F1: the scalar set of the reference's tests/tests.py:19-62
    p1 = dict(gl_sh=0.01, gl_wv=0.01, g_cuticular=1e-5, tmin_close=-8,
              tmin_open=8, vpd_open=650, vpd_close=3000, rbl_min=60,
              rbl_max=90, csl=2.4e-3, beta=250)
    d1 = dict(lw_net_day=-50, lw_net_night=-30, sw_rad_day=150, sw_rad_night=0,
              sw_albedo=0.3, temp_day=293, temp_night=290, temp_annual=285,
              tmin=285, vpd_day=1000, vpd_night=500, pressure=100e3,
              fpar=0.5, lai=1.5)
    m1 = MOD16(p1)
    drv1 = [d1[k] for k in oracle.DRIVER_NAMES]
    day, night = m1.evapotranspiration(*drv1)
    sep = pack_sep(m1.evapotranspiration(*drv1, separate=True))
    pv = [p1[k] for k in names]
    f1 = dict(params=np.array(pv, float), drivers=np.array(drv1, float),
              day=day, night=night,
              et_static=MOD16._et(pv, *drv1),
              et_static_daynight=np.array(MOD16._evapotranspiration(pv, *drv1)),
              **sep)
    # component known answers with the tests.py inputs (tests.py:92-141)
    temp_k, vpd, lai, fpar = 273.15 + 30, 1000, 1.5, 0.5
    pressure, tmin, rad = 100e3, 285, 5000
    r_corr = (101300 / pressure) * (temp_k / 293.15)**1.75
    f1['kat_evaporation_soil'] = m1.evaporation_soil(
        pressure, temp_k, vpd, fpar, rad, r_corr)
    f1['kat_transpiration_day'] = m1.transpiration(
        pressure, temp_k, vpd, lai, fpar, rad, tmin, r_corr, daytime=True)
    f1['kat_transpiration_night'] = m1.transpiration(
        pressure, temp_k, vpd, lai, fpar, rad, tmin, r_corr, daytime=False)
    f1['kat_wet_canopy'] = m1.evaporation_wet_canopy(
        pressure, temp_k, vpd, lai, fpar, rad)
    np.savez(os.path.join(HERE, 'f1_tests_scalars.npz'), **f1)

    # ---- F2: the 3-pixel set of tests/verification/verify.py:40-71 (PFT 7)
    p2 = {k: bplut[k][7] for k in names}
    p2['beta'] = 250
    tday = np.array((286.20189, 292.52667, 298.3286))
    drv2 = [-117, -65, 419, 0, 0.116, tday, tday, 289.74402,
            np.array((278.92, 284.43, 289.88)),
            np.array((710.9, 1249.4, 1979.)), np.array((710.9, 1249.4, 1979.)),
            np.array((92753.47, 92753.47, 92753.47)), 0.35839,
            np.array((0.3, 0.6, 1.0))]
    m2 = MOD16(p2)
    sep = pack_sep(m2.evapotranspiration(
        *drv2, f_wet=np.array((0, 0.4, 0.8)), separate=True))
    np.savez(os.path.join(HERE, 'f2_verify_3pixel.npz'),
             params=np.array([p2[k] for k in names], float),
             **{'drv_%s' % k: np.asarray(v, float)
                for k, v in zip(oracle.DRIVER_NAMES, drv2)}, **sep)

    # ---- F3 / F5: random 64x64 multi-class raster (f64, f32)
    for tag, dtype in (('f3_random64_f64', np.float64),
                       ('f5_random64_f32', np.float32)):
        cls, drv = synth.drivers((64, 64), seed=0, dtype=dtype)
        params = {k: bplut[k][cls] for k in names}   # notebook cell 32 idiom
        if dtype == np.float32:
            # all-f32 inputs keep numpy in float32
            params = {k: v.astype(np.float32) for k, v in params.items()}
        m3 = MOD16(params)
        day, night = m3.evapotranspiration(*drv)
        sep = pack_sep(m3.evapotranspiration(*drv, separate=True))
        assert day.dtype == dtype, day.dtype
        np.savez_compressed(
            os.path.join(HERE, tag + '.npz'), cls=cls, table=table,
            drivers=np.stack(drv), day=day, night=night, **sep)

    # ---- F4: edge cases (PFT-7 params, beta 250), one pixel per row
    base = dict(d1, fpar=0.5, lai=1.5, pressure=1e5)
    edits = [
        ('baseline', {}),
        ('lai_zero', dict(lai=0.0)),
        ('fpar_one', dict(fpar=1.0)),
        ('fpar_zero', dict(fpar=0.0)),
        ('vpd_zero', dict(vpd_day=0.0, vpd_night=0.0)),
        ('vpd_negative', dict(vpd_day=-100.0, vpd_night=-100.0)),
        ('vpd_huge', dict(vpd_day=9000.0, vpd_night=9000.0)),
        ('nan_temp_day', dict(temp_day=np.nan)),
        ('nan_temp_night', dict(temp_night=np.nan)),
        ('nan_lai', dict(lai=np.nan)),
        ('nan_fpar', dict(fpar=np.nan)),
        ('nan_vpd_day', dict(vpd_day=np.nan)),
        ('nan_pressure', dict(pressure=np.nan)),
        ('nan_albedo', dict(sw_albedo=np.nan)),
        ('nan_tmin', dict(tmin=np.nan)),
        ('nan_temp_annual', dict(temp_annual=np.nan)),
        ('g_on', dict(temp_day=296.0, temp_night=290.0)),
        ('g_on_exact5', dict(temp_day=295.0, temp_night=290.0)),
        ('g_capped', dict(temp_day=300.0, temp_night=280.0, sw_rad_day=20.0)),
        ('g_tann_hot', dict(temp_day=296.0, temp_night=290.0,
                            temp_annual=298.15)),
        ('g_tann_cold', dict(temp_day=296.0, temp_night=290.0,
                             temp_annual=265.0)),
        ('g_tann_edge', dict(temp_day=296.0, temp_night=290.0,
                             temp_annual=265.15)),
        ('rad_negative', dict(sw_rad_day=10.0, lw_net_day=-90.0)),
        ('pressure_zero', dict(pressure=0.0)),
        ('tmin_below_close', dict(tmin=260.0)),
        ('tmin_mid_ramp', dict(tmin=273.15)),
        ('vpd_below_open', dict(vpd_day=600.0)),
        ('vpd_at_open', dict(vpd_day=650.0)),
        ('vpd_at_close', dict(vpd_day=4400.0, temp_day=305.0)),
        ('vpd_above_close', dict(vpd_day=4500.0, temp_day=305.0)),
        ('rh_just_wet', dict(temp_day=293.0, vpd_day=700.0)),
        ('rh_just_dry', dict(temp_day=293.0, vpd_day=703.0)),
        ('sw_night_nonzero', dict(sw_rad_night=5.0)),
        ('cold', dict(temp_day=240.0, temp_night=235.0, tmin=233.0,
                      vpd_day=20.0, vpd_night=10.0)),
        ('hot', dict(temp_day=318.0, temp_night=305.0, tmin=303.0,
                     vpd_day=6000.0, vpd_night=2500.0)),
        ('inf_vpd', dict(vpd_day=np.inf)),
        ('zero_temp', dict(temp_day=0.0)),
    ]
    rows = []
    for _, e in edits:
        d = dict(base)
        d.update(e)
        rows.append([d[k] for k in oracle.DRIVER_NAMES])
    drv4 = [np.array(c, float) for c in zip(*rows)]
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sep = pack_sep(m2.evapotranspiration(*drv4, separate=True))
            day, night = m2.evapotranspiration(*drv4)
            # invalid-but-in-range classes give NaN params -> NaN
            cls4 = np.array([7, 0, 11, 12, 1], np.uint8)
            drv4c = [np.full(5, base[k], float) for k in oracle.DRIVER_NAMES]
            mc = MOD16({k: bplut[k][cls4] for k in names})
            dayc, nightc = mc.evapotranspiration(*drv4c)
    np.savez(os.path.join(HERE, 'f4_edge_cases.npz'),
             names=np.array([n for n, _ in edits]),
             params=np.array([p2[k] for k in names], float),
             drivers=np.stack(drv4), day=day, night=night,
             cls_case_cls=cls4, cls_case_table=table,
             cls_case_drivers=np.stack(drv4c),
             cls_case_day=dayc, cls_case_night=nightc, **sep)

    # ---- F6: per-sub-method vectors (random 256 pixels, PFT-7 params)
    _, drv = synth.drivers((256,), seed=6, special=False)
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin_, vpd_d, vpd_n, pa,
     fpar_, lai_) = drv
    f6 = dict(params=np.array([p2[k] for k in names], float),
              drivers=np.stack(drv))
    f6['svp'] = ref.svp(t_d)
    f6['svp_slope'] = ref.svp_slope(t_d)
    f6['lhv'] = ref.latent_heat_vaporization(t_d)
    f6['psychrometric_constant'] = ref.psychrometric_constant(pa, t_d)
    f6['rhumidity'] = MOD16.rhumidity(t_d, vpd_d)
    f6['air_density'] = MOD16.air_density(t_d, pa, f6['rhumidity'])
    g = m2.soil_heat_flux(sw_d * (1 - alb) + lw_d, lw_n, t_d, t_n, t_a)
    f6['soil_heat_flux_day'], f6['soil_heat_flux_night'] = g
    rs = m2.radiation_soil(lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, fpar_)
    f6['radiation_soil_day'], f6['radiation_soil_night'] = rs
    f6['surface_conductance'] = m2.surface_conductance(tmin_, vpd_d)
    rad_c = fpar_ * (sw_d * (1 - alb) + lw_d)
    f6['rad_canopy'] = rad_c
    f6['evaporation_wet_canopy'] = m2.evaporation_wet_canopy(
        pa, t_d, vpd_d, lai_, fpar_, rad_c)
    f6['evaporation_soil'] = m2.evaporation_soil(
        pa, t_d, vpd_d, fpar_, rs[0])
    ps = MOD16.potential_soil_evaporation(
        pa, t_d, vpd_d, fpar_, rs[0], vpd_open=p2['vpd_open'],
        vpd_close=p2['vpd_close'], rbl_min=p2['rbl_min'],
        rbl_max=p2['rbl_max'])
    f6['potential_soil_sat'], f6['potential_soil_unsat'] = ps
    f6['transpiration_day'] = m2.transpiration(
        pa, t_d, vpd_d, lai_, fpar_, rad_c, tmin_, daytime=True)
    f6['transpiration_night'] = m2.transpiration(
        pa, t_n, vpd_n, lai_, fpar_, fpar_ * lw_n, tmin_, daytime=False)
    np.savez_compressed(os.path.join(HERE, 'f6_submethods.npz'), **f6)

    # ---- F7: the vectorised calibration path MOD16._evapotranspiration / _et
    #      (mod16/__init__.py:162-382): (T x N) drivers, (1 x N) parameters
    T_, N_ = 12, 40
    cls7, drv7 = synth.drivers((T_, N_), seed=7, special=False)
    site_cls = cls7[0]
    pars7 = [bplut[k][site_cls].reshape(1, N_) for k in names]
    drv7[7] = np.broadcast_to(drv7[7][0], (T_, N_)).copy()       # temp_annual per site
    drv7[12][3, :5] = 0.0                                        # fpar / lai specials
    drv7[13][4, :5] = 0.0
    day7, night7 = MOD16._evapotranspiration(pars7, *drv7)
    et7 = MOD16._et(pars7, *drv7)
    rcl = [np.full((T_, N_), 1.1), np.full((T_, N_), 1.05)]
    dayr, nightr = MOD16._evapotranspiration(pars7, *drv7, r_corr_list=rcl)
    # a case where no pixel has g_surf > 0 (tmin below tmin_close everywhere)
    drv_cold = [d.copy() for d in drv7]
    drv_cold[8] = np.full((T_, N_), 240.0)
    dayc, nightc = MOD16._evapotranspiration(pars7, *drv_cold)
    np.savez_compressed(
        os.path.join(HERE, 'f7_static_path.npz'), params=np.concatenate(pars7, 0),
        drivers=np.stack(drv7), day=day7, night=night7, et=et7,
        r_corr_day=rcl[0], r_corr_night=rcl[1], day_rcorr=dayr, night_rcorr=nightr,
        tmin_cold=drv_cold[8], day_cold=dayc, night_cold=nightc)

    # ---- F8: raw drivers through the reference's own pre-processing
    #      (mod16/calibration.py:380-423: MOD16.vpd, night VPD clamp,
    #      MOD16.air_pressure, fPAR / 100, LAI / 10) and its forward run, plus
    #      the 8-day product unit of tests/verification/verify2.py:113-115
    rng = np.random.default_rng(8)
    shp = (48, 50)
    cls8, drv8 = synth.drivers(shp, seed=8, special=False)
    qv_d, qv_n = rng.uniform(0.001, 0.02, shp), rng.uniform(0.001, 0.02, shp)
    ps_d, ps_n = rng.uniform(70000, 101340, shp), rng.uniform(70000, 101340, shp)
    elev = rng.uniform(-50, 3500, shp)
    fpar_pct = rng.integers(0, 101, shp).astype(np.uint8)
    lai_x10 = rng.integers(0, 71, shp).astype(np.uint8)
    fpar_pct[0, :4] = (249, 250, 255, 0)
    lai_x10[1, :4] = (249, 253, 255, 0)
    hours = rng.uniform(8, 16, shp)
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin8) = drv8[:9]
    vpd_d = MOD16.vpd(qv_d, ps_d, t_d)
    vpd_n = MOD16.vpd(qv_n, ps_n, t_n)
    vpd_n = np.where(vpd_n < 0, 0, vpd_n)
    pa8 = MOD16.air_pressure(elev)
    fpar8 = np.where(fpar_pct >= 249, np.nan, fpar_pct.astype(np.float64))
    lai8 = np.where(lai_x10 >= 249, np.nan, lai_x10.astype(np.float64))
    fpar8 /= 100
    lai8 /= 10
    m8 = MOD16({k: bplut[k][cls8] for k in names})
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        day8, night8 = m8.evapotranspiration(
            lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin8, vpd_d, vpd_n, pa8, fpar8, lai8)
    total8 = ((day8 * hours * 8 * 60 * 60) + (night8 * (24 - hours) * 8 * 60 * 60))
    np.savez_compressed(
        os.path.join(HERE, 'f8_raw_drivers.npz'), cls=cls8, table=table,
        raw=np.stack([lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin8, qv_d, qv_n, ps_d,
                      ps_n, elev]),
        fpar_pct=fpar_pct, lai_x10=lai_x10, day_hours=hours, vpd_day=vpd_d, vpd_night=vpd_n,
        pressure=pa8, day=day8, night=night8, total8=total8)

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith('.npz'):
            print('%-28s %7d bytes' % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == '__main__':
    main()
