"""ctypes binding of libmara_hip.so (C ABI: include/mara_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MARA_HIP_LIBRARY: another build of the same library (A/B measurements of kernel variants on one box)
_LIB_PATH = os.environ.get("MARA_HIP_LIBRARY") or os.path.join(_HERE, "libmara_hip.so")

# enums of include/mara_hip.h
OK = 0
RIEMANN_HLLE, RIEMANN_HLLC = 0, 1
BC_OUTFLOW, BC_PERIODIC, BC_EXTERNAL, BC_REFLECT, BC_INFLOW = 0, 1, 2, 3, 4
ARITH_STRICT, ARITH_FAST = 0, 1
STATUS_NEG_DENSITY, STATUS_NEG_PRESSURE, STATUS_C2P_FAILED, STATUS_NAN = 1, 2, 4, 8


class MaraHipError(RuntimeError):
    pass


class EulerCartDesc(C.Structure):
    _fields_ = [
        ("rank", C.c_int),
        ("n", C.c_int * 3),
        ("dl", C.c_double * 3),
        ("gamma", C.c_double),
        ("plm_theta", C.c_double),
        ("riemann", C.c_int),
        ("bc_lo0", C.c_int),
        ("bc_hi0", C.c_int),
        ("bc_transverse", C.c_int),
        ("arith", C.c_int),
        ("chunk_rows", C.c_int),
        ("tail_rows", C.c_int),
        ("tail_chunk_rows", C.c_int),
        ("fuse_stages", C.c_int),
        ("planar", C.c_int),
    ]


class StepResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("reserved", C.c_int32), ("first_bad_index", C.c_uint64)]


class SedovDesc(C.Structure):
    _fields_ = [("nz", C.c_int), ("gamma", C.c_double), ("system", C.c_int), ("arith", C.c_int)]


class CloudDesc(C.Structure):
    _fields_ = [("nr", C.c_int), ("nq", C.c_int), ("nr_global", C.c_int), ("row_offset", C.c_int), ("gamma", C.c_double),
                ("plm_theta", C.c_double), ("temperature_floor", C.c_double), ("bc_lo0", C.c_int), ("bc_hi0", C.c_int),
                ("arith", C.c_int), ("chunk_rows", C.c_int), ("tail_rows", C.c_int), ("tail_chunk_rows", C.c_int), ("fuse_stages", C.c_int), ("planar", C.c_int)]


class SlabPlanMsg(C.Structure):
    _fields_ = [("send", C.c_int), ("peer", C.c_int), ("first_row", C.c_int), ("rows", C.c_int)]


class SlabPlan(C.Structure):
    """mh_slab_plan (include/mara_hip.h): the device-free decisions of a slab decomposition, made in csrc/slab_plan.hpp"""
    _fields_ = [("row0", C.c_int), ("row1", C.c_int), ("lo", C.c_int), ("hi", C.c_int), ("ghost_rows", C.c_int), ("edge_rows", C.c_int),
                ("exchanges_per_step", C.c_int), ("nmsg", C.c_int), ("msg", SlabPlanMsg * 4)]


class BinaryDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("block_size", C.c_int32), ("domain_radius", C.c_double), ("mach_number", C.c_double),
                ("alpha", C.c_double), ("nu", C.c_double), ("alpha_cutoff_radius", C.c_double), ("sink_rate", C.c_double),
                ("sink_radius", C.c_double), ("softening_radius", C.c_double), ("density_floor", C.c_double),
                ("plm_theta", C.c_double), ("axisymmetric_cs2", C.c_int32), ("chunk_rows", C.c_int32), ("angmom_form", C.c_int32),
                ("arith", C.c_int32), ("gst_suppr_radius", C.c_double)]


class OrbitalElements(C.Structure):
    _fields_ = [("separation", C.c_double), ("total_mass", C.c_double), ("mass_ratio", C.c_double), ("eccentricity", C.c_double)]


class FullOrbitalElements(C.Structure):
    _fields_ = [("pomega", C.c_double), ("tau", C.c_double), ("cm_position_x", C.c_double), ("cm_position_y", C.c_double),
                ("cm_velocity_x", C.c_double), ("cm_velocity_y", C.c_double), ("elements", OrbitalElements)]

    def as_array(self):
        e = self.elements
        return [self.pomega, self.tau, self.cm_position_x, self.cm_position_y, self.cm_velocity_x, self.cm_velocity_y,
                e.separation, e.total_mass, e.mass_ratio, e.eccentricity]


class BinaryState(C.Structure):
    _fields_ = [("time", C.c_double), ("iteration", C.c_int64), ("mass_accreted_on", C.c_double * 2),
                ("angular_momentum_accreted_on", C.c_double * 2), ("integrated_torque_on", C.c_double * 2),
                ("work_done_on", C.c_double * 2), ("mass_ejected", C.c_double), ("angular_momentum_ejected", C.c_double),
                ("orbital_elements_acc", FullOrbitalElements), ("orbital_elements_grav", FullOrbitalElements),
                ("orbital_elements", FullOrbitalElements)]


class BinaryModel(C.Structure):
    _fields_ = [("softening_radius", C.c_double), ("disk_radius", C.c_double), ("mach_number", C.c_double),
                ("disk_mass", C.c_double), ("ambient_density", C.c_double), ("mdot", C.c_double), ("counter_rotate", C.c_int32),
                ("angmom_form", C.c_int32), ("buffer_damping_rate", C.c_double), ("domain_radius", C.c_double), ("cfl_number", C.c_double)]


class TreeBlock(C.Structure):
    _fields_ = [("level", C.c_int32), ("i", C.c_int32), ("j", C.c_int32)]


class BinaryRun(C.Structure):
    _fields_ = [("rk_order", C.c_int32), ("fixed_dt", C.c_int32), ("no_accretion_force", C.c_int32), ("reserved", C.c_int32),
                ("cfl_number", C.c_double), ("recommended_time_step", C.c_double), ("begin_live_binary", C.c_double)]


BINARY_NTOTALS = 18
T_MASS_ACC, T_L_ACC, T_TORQUE, T_PX_ACC, T_PY_ACC, T_FX, T_FY, T_WORK, T_MASS_EJ, T_L_EJ = 0, 2, 4, 6, 8, 10, 12, 14, 16, 17

# every symbol include/mara_hip.h declares: (name, restype, argtypes)
_vp, _dp, _sz, _i, _d = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double
_descp = C.POINTER(EulerCartDesc)
SYMBOLS = [
    ("mh_euler_cart_field_doubles", _sz, [_descp]),
    ("mh_euler_cart_stage", _i, [_descp, _dp, _dp, _dp, _d, _d, _i, _i, _vp, _vp]),
    ("mh_euler_cart_fill_ghosts", _i, [_descp, _dp, _vp]),
    ("mh_aos_to_soa", _i, [_dp, _dp, _i, _i, _sz, _vp]),
    ("mh_soa_to_aos", _i, [_dp, _dp, _i, _i, _sz, _vp]),
    ("mh_calib_stream_copy", _i, [_dp, _dp, _sz, _vp]),
    ("mh_create", _i, [C.POINTER(_vp), _i]),
    ("mh_destroy", None, [_vp]),
    ("mh_last_error", C.c_char_p, [_vp]),
    ("mh_euler_cart_configure", _i, [_vp, _descp, _i]),
    ("mh_sedov_configure", _i, [_vp, C.POINTER(SedovDesc), _vp]),
    ("mh_cloud_geometry_doubles", _sz, [C.POINTER(CloudDesc)]),
    ("mh_cloud_pack_geometry", _i, [C.POINTER(CloudDesc), _vp, _vp, _vp]),
    ("mh_cloud_stage", _i, [C.POINTER(CloudDesc), _dp, _dp, _dp, _dp, _dp, _d, _d, _i, _i, _vp, _vp]),
    ("mh_cloud_configure", _i, [_vp, C.POINTER(CloudDesc), _vp, _vp, _i]),
    ("mh_cloud_set_inflow", _i, [_vp, _vp]),
    ("mh_cloud_diagnostics", _i, [_vp, _vp, _vp, _vp]),
    ("mh_sedov_diagnostics", _i, [_vp, _vp, _vp]),
    ("mh_upload", _i, [_vp, _vp, _sz]),
    ("mh_download", _i, [_vp, _vp, _sz]),
    ("mh_step", _i, [_vp, _d, _i]),
    ("mh_synchronize", _i, [_vp]),
    ("mh_status_word", _i, [_vp, C.POINTER(C.c_int32)]),
    ("mh_status", _i, [_vp, C.POINTER(StepResult)]),
    ("mh_step_checked", _i, [_vp, _d, C.POINTER(StepResult)]),
    ("mh_field_ptr", _vp, [_vp, _i]),
    ("mh_profile_enable", _i, [_vp, _i]),
    ("mh_profile_read", _i, [_vp, C.POINTER(_d), C.POINTER(_i)]),
    ("mh_comm_unique_id", _i, [_vp]),
    ("mh_slab_create", _i, [C.POINTER(_vp), _descp, _i, _i, _i, _vp, _i, _i]),
    ("mh_slab_connect", _i, [_vp, _vp]),
    ("mh_slab_use_comm", _i, [_vp, _vp]),
    ("mh_comm_create", _i, [C.POINTER(_vp), _vp, _i, _i, _i]),
    ("mh_comm_destroy", None, [_vp]),
    ("mh_slab_cloud_create", _i, [C.POINTER(_vp), C.POINTER(CloudDesc), _vp, _vp, _i, _i, _i, _vp, _i]),
    ("mh_slab_launches_per_step", _i, [_vp]),
    ("mh_slab_plan_make", _i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(SlabPlan)]),
    ("mh_slab_plan_of", _i, [_vp, C.POINTER(SlabPlan)]),
    ("mh_slab_set_inflow", _i, [_vp, _vp]),
    ("mh_slab_group_set_inflow", _i, [C.POINTER(_vp), _i, _vp]),
    ("mh_slab_group_create", _i, [C.POINTER(_vp), _descp, _i, _i, _i]),
    ("mh_slab_cloud_group_create", _i, [C.POINTER(_vp), C.POINTER(CloudDesc), _vp, _vp, _i, _i, _i]),
    ("mh_slab_group_create_on", _i, [C.POINTER(_vp), _descp, _i, _i, C.POINTER(_i)]),
    ("mh_slab_cloud_group_create_on", _i, [C.POINTER(_vp), C.POINTER(CloudDesc), _vp, _vp, _i, _i, C.POINTER(_i)]),
    ("mh_slab_group_upload", _i, [C.POINTER(_vp), _i, _vp]),
    ("mh_slab_group_download", _i, [C.POINTER(_vp), _i, _vp]),
    ("mh_slab_group_step", _i, [C.POINTER(_vp), _i, _d, _i]),
    ("mh_slab_status", _i, [_vp, C.POINTER(StepResult)]),
    ("mh_slab_destroy", None, [_vp]),
    ("mh_slab_rows", _i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    ("mh_slab_upload", _i, [_vp, _vp]),
    ("mh_slab_download", _i, [_vp, _vp]),
    ("mh_slab_step", _i, [_vp, _d, _i, _i]),
    ("mh_slab_synchronize", _i, [_vp]),
    ("mh_slab_status_word", _i, [_vp, C.POINTER(C.c_int32)]),
    ("mh_slab_field_ptr", _vp, [_vp, _i]),
    ("mh_slab_profile_enable", _i, [_vp, _i]),
    ("mh_slab_profile_read", _i, [_vp, C.POINTER(_d), C.POINTER(_i), C.POINTER(_i)]),
    ("mh_block_create", _i, [C.POINTER(_vp), _descp, _i, _i, _i, _vp, _i, _i]),
    ("mh_block_connect", _i, [_vp, _vp]),
    ("mh_block_use_comm", _i, [_vp, _vp]),
    ("mh_block_destroy", None, [_vp]),
    ("mh_block_extent", _i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("mh_block_neighbours", _i, [_vp, C.POINTER(_i), C.POINTER(_sz)]),
    ("mh_block_upload", _i, [_vp, _vp]),
    ("mh_block_download", _i, [_vp, _vp]),
    ("mh_block_step", _i, [_vp, _d, _i]),
    ("mh_block_synchronize", _i, [_vp]),
    ("mh_block_status", _i, [_vp, C.POINTER(StepResult), C.POINTER(_i)]),
    ("mh_block_profile", _i, [_vp, _i, C.POINTER(_d), C.POINTER(_i), C.POINTER(C.c_long)]),
    ("mh_block_group_create", _i, [C.POINTER(_vp), _descp, _i, _i, _i]),
    ("mh_block_group_upload", _i, [C.POINTER(_vp), _i, _vp]),
    ("mh_block_group_download", _i, [C.POINTER(_vp), _i, _vp]),
    ("mh_block_group_step", _i, [C.POINTER(_vp), _i, _d, _i]),
    ("mh_plm_gradient_n", _i, [_sz, _dp, _dp, _dp, _d, _dp, _i, _vp]),
    ("mh_euler_recover_primitive_n", _i, [_sz, _dp, _d, _d, _dp, _i, _vp]),
    ("mh_euler_to_conserved_n", _i, [_sz, _dp, _d, _dp, _i, _vp]),
    ("mh_euler_riemann_n", _i, [_sz, _dp, _dp, _i, _d, _i, _dp, _i, _vp]),
    ("mh_srhd_recover_primitive_n", _i, [_sz, _dp, _d, _d, _dp, _vp, _vp]),
    ("mh_srhd_to_conserved_n", _i, [_sz, _dp, _d, _dp, _vp]),
    ("mh_srhd_riemann_hlle_n", _i, [_sz, _dp, _dp, _i, _d, _dp, _vp]),
    ("mh_srhd_source_terms_n", _i, [_sz, _dp, _dp, _dp, _d, _dp, _vp]),
    ("mh_iso2d_to_conserved_n", _i, [_sz, _dp, _dp, _vp]),
    ("mh_iso2d_recover_primitive_n", _i, [_sz, _dp, _dp, _vp, _vp]),
    ("mh_iso2d_to_conserved_angmom_n", _i, [_sz, _dp, _dp, _dp, _vp]),
    ("mh_iso2d_recover_primitive_angmom_n", _i, [_sz, _dp, _dp, _dp, _vp, _vp]),
    ("mh_iso2d_flux_n", _i, [_sz, _dp, _dp, _i, _dp, _vp]),
    ("mh_iso2d_wavespeeds_n", _i, [_sz, _dp, _dp, _i, _dp, _vp]),
    ("mh_iso2d_riemann_n", _i, [_sz, _dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _vp, _vp]),
    ("mh_partition_rows", None, [_sz, _sz, _sz, C.POINTER(_sz), C.POINTER(_sz)]),
    ("mh_propose_block_decomposition", _i, [_i, C.c_ulong, C.POINTER(C.c_ulong)]),
    ("mh_block_layout", _i, [C.POINTER(_i), _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("mh_two_body_state", _i, [_vp, _d, _vp]),
    ("mh_orbital_elements_from_state", _i, [_vp, _d, _vp]),
    ("mh_orbital_elements_diff", None, [_vp, _vp, _vp]),
    ("mh_binary_field_doubles", _sz, [C.POINTER(BinaryDesc)]),
    ("mh_binary_scratch_doubles", _sz, [C.POINTER(BinaryDesc)]),
    ("mh_binary_stage", _i, [C.POINTER(BinaryDesc), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _vp, _d, _d, _dp, _dp, _vp, _vp]),
    ("mh_binary_max_wavespeed", _i, [C.POINTER(BinaryDesc), _dp, _dp, _dp, _vp, _dp, _vp]),
    ("mh_binary_vertices", _i, [_i, _i, _d, _vp]),
    ("mh_binary_solver_data", _i, [C.POINTER(BinaryModel), _i, _vp, _vp, _vp, _vp, C.POINTER(_d)]),
    ("mh_binary_tree_build", _i, [_i, _i, _d, _d, _vp, _i]),
    ("mh_binary_tree_vertices", _i, [_i, _d, _vp, _i, _vp]),
    ("mh_binary_tree_solver_data", _i, [C.POINTER(BinaryModel), _i, _vp, _i, _vp, _vp, _vp, C.POINTER(_d)]),
    ("mh_binary_tree_create", _i, [C.POINTER(_vp), _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _i, _vp, _vp, _vp]),
    ("mh_binary_tree_curve_order", _i, [_vp, _i, _vp]),
    ("mh_binary_tree_band_create", _i, [C.POINTER(_vp), _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    ("mh_binary_tree_group_create", _i, [_vp, _i, _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _i, _vp, _vp, _vp]),
    ("mh_binary_tree_owned_blocks", _i, [_vp, _vp, C.POINTER(_i)]),
    ("mh_binary_create", _i, [C.POINTER(_vp), _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _vp, _vp, _vp]),
    ("mh_binary_destroy", None, [_vp]),
    ("mh_binary_set_solution", _i, [_vp, _vp, C.POINTER(BinaryState)]),
    ("mh_binary_get_solution", _i, [_vp, _vp, C.POINTER(BinaryState)]),
    ("mh_binary_next", _i, [_vp, _i, C.POINTER(_i)]),
    ("mh_binary_band_create", _i, [C.POINTER(_vp), _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _vp, _vp, _vp, _i, _i, _vp, _i]),
    ("mh_binary_band_rows", _i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    ("mh_binary_band_use_comm", _i, [_vp, _vp]),
    ("mh_binary_band_set_edge_rows", _i, [_vp, _i]),
    ("mh_binary_last_failure", _i, [_vp, C.POINTER(StepResult)]),
    ("mh_binary_group_create", _i, [C.POINTER(_vp), _i, _i, C.POINTER(BinaryDesc), C.POINTER(BinaryRun), _vp, _vp, _vp, _vp]),
    ("mh_binary_group_set_solution", _i, [C.POINTER(_vp), _i, _vp, C.POINTER(BinaryState)]),
    ("mh_binary_group_get_solution", _i, [C.POINTER(_vp), _i, _vp, C.POINTER(BinaryState)]),
    ("mh_binary_group_next", _i, [C.POINTER(_vp), _i, _i, C.POINTER(_i)]),
    ("mh_binary_last_dt", _d, [_vp]),
    ("mh_binary_field_ptr", _vp, [_vp]),
    ("mh_binary_profile", _i, [_vp, _i, C.POINTER(_d), C.POINTER(_i)]),
    ("mh_binary_disk_totals", _i, [_vp, C.POINTER(_d), C.POINTER(_d)]),
    ("mh_binary_diagnostic_fields", _i, [_vp, _vp, _vp, _vp]),
    ("mh_device_count", _i, []),
    ("mh_device_cu_count", _i, []),
    ("mh_debug_last_fused_cut", _i, [_i, C.POINTER(C.c_int32)]),
    ("mh_malloc", _i, [C.POINTER(_vp), _sz]),
    ("mh_free", _i, [_vp]),
    ("mh_memcpy_h2d", _i, [_vp, _vp, _sz]),
    ("mh_memcpy_d2h", _i, [_vp, _vp, _sz]),
    ("mh_device_synchronize", _i, []),
    ("mh_debug_row_range", _i, [_i, C.POINTER(C.c_int32), _i]),
    ("mh_field_is_planar", _i, [_vp]),
    ("mh_slab_is_planar", _i, [_vp]),
]

_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Load libmara_hip.so. Raises MaraHipError if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MaraHipError(
            "libmara_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mara3_amd/csrc`. mara3_amd has no CPU fallback." % _LIB_PATH)
    # Inside a Python process that also uses torch, load torch FIRST: both link the HIP runtime by the same soname
    # (libamdhip64.so.7) and whichever is loaded first serves the whole process; torch must get the copy it ships with.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as e:
        raise MaraHipError("cannot load %s: %s" % (_LIB_PATH, e))
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise MaraHipError("libmara_hip.so does not export %s" % name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != OK:
        msg = load_library().mh_last_error(ctx)
        raise MaraHipError("mara_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
