"""ctypes mirrors of the C structs in include/xrt_hip.h (layout checked against
xrt_hip_sizeof at load time in hipcalls)."""
import ctypes

MAX_ROT = 8
BUCKETS, BUCKET_SHIFT, BUCKET_KEY0 = 1280, 46, 0x3FF << 6
MAX_ELEM = 4
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)

SURF_FLAT, SURF_TOROID, SURF_BENTFLAT, SURF_BLAZED, SURF_ELLIPSE_PARAM = 0, 1, 2, 3, 4
SURF_PARABOLOID, SURF_CONE, SURF_SAGITTAL, SURF_BENT_BRAGG = 5, 6, 7, 8
SURF_VFM, SURF_DUALVFM, SURF_DICED, SURF_USER = 9, 10, 11, 12
SHAPE_RECT, SHAPE_ROUND, SHAPE_POLYGON = 0, 1, 2
OVER_XMIN, OVER_XMAX, OVER_YMIN, OVER_YMAX = 1, 2, 4, 8
MAT_NONE, MAT_MIRROR, MAT_THIN_MIRROR, MAT_PLATE, MAT_CRYSTAL, MAT_MULTILAYER = 0, 1, 2, 3, 4, 5


class Beam(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int64)] + \
        [(f, ctypes.c_void_p) for f in
         ('x', 'y', 'z', 'a', 'b', 'c', 'path', 'E', 'Jss', 'Jpp', 'Jsp_ri',
          'state', 'Es_ri', 'Ep_ri')]


class Rotation(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int32),
                ('axis', ctypes.c_int32 * MAX_ROT),
                ('cosa', ctypes.c_double * MAX_ROT),
                ('sina', ctypes.c_double * MAX_ROT)]


class Pass(ctypes.Structure):
    _fields_ = [
        ('good_mode', ctypes.c_int32),
        ('in_is_global', ctypes.c_int32),
        ('center', ctypes.c_double * 3),
        ('sin_az', ctypes.c_double),
        ('cos_az', ctypes.c_double),
        ('to_local', Rotation),
        ('to_virgin', Rotation),
        ('shift', ctypes.c_double * 3),
        ('invert_normal', ctypes.c_int32),
        ('no_intersection_search', ctypes.c_int32),
        ('surf_kind', ctypes.c_int32),
        ('surf_p', ctypes.c_double * 12),
        ('n_const', ctypes.c_double * 6),
        ('asymmetric', ctypes.c_int32),
        ('shape', ctypes.c_int32),
        ('phys_x', ctypes.c_double * 2),
        ('phys_y', ctypes.c_double * 2),
        ('has_opt_x', ctypes.c_int32),
        ('has_opt_y', ctypes.c_int32),
        ('opt_x', ctypes.c_double * 2),
        ('opt_y', ctypes.c_double * 2),
        ('over_mask', ctypes.c_int32),
        ('lost_num', ctypes.c_int32),
        ('roll', ctypes.c_double),
        ('cos_roll', ctypes.c_double),
        ('sin_roll', ctypes.c_double),
        ('out_to_global', ctypes.c_int32),
        ('only_state1_out', ctypes.c_int32),
        ('zero_local_not_entering', ctypes.c_int32),
        ('force_lost_out', ctypes.c_int32),
        ('grating', ctypes.c_int32),
        ('grating_axis', ctypes.c_int32),
        ('grating_order', ctypes.c_int32),
        ('g_ncoef', ctypes.c_int32),
        ('g_rho0', ctypes.c_double),
        ('g_coef', ctypes.c_double * 8),
        ('g_const', ctypes.c_double * 3),
        ('poly_n', ctypes.c_int32),
        ('poly_xy', ctypes.c_void_p),
        ('order_ray', ctypes.c_void_p),
        ('zone_n', ctypes.c_int32),
        ('zone_black', ctypes.c_int32),
        ('zone_r', ctypes.c_void_p),
        ('eff_n', ctypes.c_int32),
        ('eff_order', ctypes.c_int32 * 8),
        ('eff_amp', ctypes.c_double * 8),
        ('state_ray', ctypes.c_void_p),
        ('g_ray_x', ctypes.c_void_p),
        ('g_ray_y', ctypes.c_void_p),
        ('method_hint', ctypes.c_void_p),
        ('user_unit', ctypes.c_void_p),
        ('eff_tab_n', ctypes.c_int32),
        ('eff_tab_E', ctypes.c_void_p),
        ('eff_tab_I', ctypes.c_void_p),
        ('fe_ntx', ctypes.c_int32),
        ('fe_nty', ctypes.c_int32),
        ('fe_k', ctypes.c_int32),
        ('fe_reserved', ctypes.c_int32),
        ('fe_tx', ctypes.c_void_p),
        ('fe_ty', ctypes.c_void_p),
        ('fe_c', ctypes.c_void_p),
        ('fe_cx', ctypes.c_void_p),
        ('fe_cy', ctypes.c_void_p),
        ('fe_shift', ctypes.c_double * 2),
        ('fe_grid', ctypes.c_int32 * 2),
        ('fe_lo', ctypes.c_double * 2),
        ('fe_step', ctypes.c_double * 2),
        ('fe_hi', ctypes.c_double * 2),
        ('fe_inv', (ctypes.c_double * 3) * 2),
        ('is_multi', ctypes.c_int32),
        ('need_elevation_map', ctypes.c_int32),
    ]


class Material(ctypes.Structure):
    _fields_ = [
        ('kind', ctypes.c_int32),
        ('from_vacuum', ctypes.c_int32),
        ('nelem', ctypes.c_int32),
        ('Z', ctypes.c_int32 * MAX_ELEM),
        ('tab_n', ctypes.c_int32 * MAX_ELEM),
        ('quantity', ctypes.c_double * MAX_ELEM),
        ('tab_E', ctypes.c_void_p * MAX_ELEM),
        ('tab_f1', ctypes.c_void_p * MAX_ELEM),
        ('tab_f2', ctypes.c_void_p * MAX_ELEM),
        ('tab_bucket', ctypes.c_void_p * MAX_ELEM),
        ('f0_hkl', ctypes.c_double),
        ('d2f_re', ctypes.c_double),
        ('d2f_im', ctypes.c_double),
        ('rho', ctypes.c_double),
        ('mass', ctypes.c_double),
        ('t', ctypes.c_double),
        ('structure', ctypes.c_int32),
        ('hkl', ctypes.c_int32 * 3),
        ('geom_bragg', ctypes.c_int32),
        ('geom_transmitted', ctypes.c_int32),
        ('thick', ctypes.c_int32),
        ('d', ctypes.c_double),
        ('chi_to_f', ctypes.c_double),
        ('fact_dw', ctypes.c_double),
        ('t_crystal', ctypes.c_double),
        ('layers', ctypes.c_void_p),
        ('cell', ctypes.c_void_p),
        ('n_fixed', ctypes.c_int32),
        ('n_re', ctypes.c_double),
        ('n_im', ctypes.c_double),
        ('n_ray', ctypes.c_void_p),
    ]


class Cell(ctypes.Structure):
    _fields_ = [('w', ctypes.c_double * 4), ('f0', ctypes.c_double * 4),
                ('s', ctypes.c_double * 2 * 4), ('sm', ctypes.c_double * 2 * 4)]


class Multilayer(ctypes.Structure):
    _fields_ = [
        ('top', Material),
        ('bottom', Material),
        ('substrate', Material),
        ('npairs', ctypes.c_int32),
        ('transmitted', ctypes.c_int32),
        ('uniform', ctypes.c_int32),
        ('dti', ctypes.c_void_p),
        ('dbi', ctypes.c_void_p),
        ('id2', ctypes.c_double),
        ('bs_rough2', ctypes.c_double),
        ('subst_thickness', ctypes.c_double),
    ]


class Screen(ctypes.Structure):
    _fields_ = [('center', ctypes.c_double * 3),
                ('ex', ctypes.c_double * 3),
                ('ey', ctypes.c_double * 3),
                ('ez', ctypes.c_double * 3),
                ('compress_x', ctypes.c_double),
                ('compress_z', ctypes.c_double),
                ('lost_num', ctypes.c_int32),
                ('only_positive_path', ctypes.c_int32),
                ('radius', ctypes.c_double),
                ('theta_offset', ctypes.c_double),
                ('phi_offset', ctypes.c_double),
                ('out_theta', ctypes.c_void_p),
                ('out_phi', ctypes.c_void_p)]


class Aperture(ctypes.Structure):
    _fields_ = [('center', ctypes.c_double * 3),
                ('ex', ctypes.c_double * 3),
                ('ey', ctypes.c_double * 3),
                ('ez', ctypes.c_double * 3),
                ('sin_az', ctypes.c_double),
                ('cos_az', ctypes.c_double),
                ('blade', ctypes.c_double * 4),
                ('blade_mask', ctypes.c_int32),
                ('is_beam_stop', ctypes.c_int32),
                ('lost_num', ctypes.c_int32),
                ('round', ctypes.c_int32),
                ('radius', ctypes.c_double),
                ('has_shade', ctypes.c_int32),
                ('glo_adds_path', ctypes.c_int32),
                ('shade', ctypes.c_double * 2),
                ('poly_n', ctypes.c_int32),
                ('own_marks', ctypes.c_int32),
                ('poly_xz', ctypes.c_void_p)]


class Undulator(ctypes.Structure):
    _fields_ = [('mode', ctypes.c_int32),
                ('nper', ctypes.c_int32),
                ('Kx', ctypes.c_double),
                ('Ky', ctypes.c_double),
                ('alpha_s', ctypes.c_double),
                ('r0z', ctypes.c_double),
                ('jend', ctypes.c_int64),
                ('tg', ctypes.c_void_p),
                ('ag', ctypes.c_void_p),
                ('sintg', ctypes.c_void_p),
                ('costg', ctypes.c_void_p),
                ('sintgph', ctypes.c_void_p),
                ('costgph', ctypes.c_void_p),
                ('workspace_packed', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


class UndulatorMap(ctypes.Structure):
    _fields_ = [('L0', ctypes.c_double),
                ('Np', ctypes.c_double),
                ('gamma0', ctypes.c_double),
                ('eI', ctypes.c_double),
                ('dstep', ctypes.c_double),
                ('harmonic', ctypes.c_double),
                ('has_harmonic', ctypes.c_int32),
                ('dist_bw', ctypes.c_int32)]


class Plot(ctypes.Structure):
    _fields_ = [('x_factor', ctypes.c_double),
                ('y_factor', ctypes.c_double),
                ('c_factor', ctypes.c_double),
                ('source_weight', ctypes.c_double),
                ('x_lim', ctypes.c_double * 2),
                ('y_lim', ctypes.c_double * 2),
                ('c_lim', ctypes.c_double * 2),
                ('color_factor', ctypes.c_double),
                ('color_saturation', ctypes.c_double),
                ('bins_x', ctypes.c_int32),
                ('bins_y', ctypes.c_int32),
                ('bins_c', ctypes.c_int32),
                ('ray_flags', ctypes.c_int32),
                ('flux_kind', ctypes.c_int32)]


# which quantity of a screen's image a plot axis shows (XRT_HIP_FIELD_*)
PLOT_FIELDS = {'x': 0, 'y': 1, 'z': 2, 'a': 3, 'b': 4, 'c': 5, 'path': 6, 'E': 7,
               'xprime': 8, 'zprime': 9}


class PlotTail(ctypes.Structure):
    """xrt_hip_plot_tail: an XYCPlot in the tail of a pass (include/xrt_hip.h)."""
    _fields_ = [('plot', Plot),
                ('x_field', ctypes.c_int32),
                ('y_field', ctypes.c_int32),
                ('c_field', ctypes.c_int32),
                ('reserved', ctypes.c_int32),
                ('hist2d', ctypes.c_void_p),
                ('hist2d_rgb', ctypes.c_void_p),
                ('hist_x', ctypes.c_void_p),
                ('hist_y', ctypes.c_void_p),
                ('hist_c', ctypes.c_void_p),
                ('counters', ctypes.c_void_p),
                ('workspace', ctypes.c_void_p),
                ('workspace_bytes', ctypes.c_size_t)]


class CustomField(ctypes.Structure):
    _fields_ = [('filament', ctypes.c_int32),
                ('near_field', ctypes.c_int32),
                ('betam', ctypes.c_double),
                ('R0', ctypes.c_double),
                ('wc', ctypes.c_double),
                ('jend', ctypes.c_int64)] + \
        [(k, ctypes.c_void_p) for k in ('tg', 'ag', 'Bx', 'By', 'Bz', 'betax',
                                        'betay', 'trajx', 'trajy', 'trajz')] + \
        [('carrier_form', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class Bend(ctypes.Structure):
    _fields_ = [('gamma', ctypes.c_double), ('B', ctypes.c_double), ('K', ctypes.c_double),
                ('poles', ctypes.c_double), ('eI', ctypes.c_double),
                ('wiggler', ctypes.c_int32), ('per_bandwidth', ctypes.c_int32)]


class Gauss(ctypes.Structure):
    _fields_ = [('w0x', ctypes.c_double), ('w0z', ctypes.c_double),
                ('astigmatic', ctypes.c_int32), ('mode', ctypes.c_int32),
                ('l', ctypes.c_int32), ('p', ctypes.c_int32), ('m', ctypes.c_int32),
                ('n', ctypes.c_int32), ('clp', ctypes.c_double)]


LAW_NONE, LAW_NORMAL, LAW_FLAT, LAW_NORMAL_UNIFORM = 0, 1, 2, 3
MAX_LINES = 16


class GeoSource(ctypes.Structure):
    _fields_ = [('seed', ctypes.c_uint64),
                ('call', ctypes.c_uint32),
                ('slopes', ctypes.c_int32),
                ('law', ctypes.c_int32 * 5),
                ('p0', ctypes.c_double * 5),
                ('p1', ctypes.c_double * 5),
                ('annulus_xz', ctypes.c_int32),
                ('annulus_ac', ctypes.c_int32),
                ('ann_xz', ctypes.c_double * 4),
                ('ann_ac', ctypes.c_double * 4),
                ('e_law', ctypes.c_int32),
                ('filament', ctypes.c_int32),
                ('n_lines', ctypes.c_int32),
                ('random_ep', ctypes.c_int32),
                ('e_p0', ctypes.c_double),
                ('e_p1', ctypes.c_double),
                ('e_lines', ctypes.c_double * MAX_LINES),
                ('e_cdf', ctypes.c_double * MAX_LINES),
                ('Jss', ctypes.c_double),
                ('Jpp', ctypes.c_double),
                ('Jsp', ctypes.c_double * 2),
                ('Es', ctypes.c_double * 2),
                ('Ep', ctypes.c_double * 2),
                ('rot', Rotation),
                ('to_global', ctypes.c_int32),
                ('state', ctypes.c_int32),
                ('sin_az', ctypes.c_double),
                ('cos_az', ctypes.c_double),
                ('center', ctypes.c_double * 3),
                ('call_dev', ctypes.c_void_p)]


class Bounce(ctypes.Structure):
    _fields_ = [
        ('nrefl_in', ctypes.c_void_p),
        ('nrefl_out', ctypes.c_void_p),
        ('theta', ctypes.c_void_p),
        ('elev_in', ctypes.c_void_p * 4),
        ('elev_out', ctypes.c_void_p * 4),
        ('spr_out', ctypes.c_void_p * 3),
        ('entering_hint', ctypes.c_int64),
        ('assume_hit_brent', ctypes.c_int32),
        ('assume_tangency_brent', ctypes.c_int32),
        ('found_host', ctypes.c_void_p),
    ]


class Tail(ctypes.Structure):
    """xrt_hip_tail: apertures, a screen and a plot in the tail of a pass (include/xrt_hip.h)."""
    _fields_ = [('n_apertures', ctypes.c_int32),
                ('keep_screen', ctypes.c_int32),
                ('aperture', Aperture * 2),
                ('screen', ctypes.c_void_p),
                ('out_screen', ctypes.c_void_p),
                ('plot', ctypes.c_void_p)]


STRUCTS = (Beam, Rotation, Pass, Material, Screen, Aperture, Undulator,
           UndulatorMap, Plot, CustomField, Bend, Multilayer, Gauss, GeoSource, Bounce, PlotTail,
           Tail)
