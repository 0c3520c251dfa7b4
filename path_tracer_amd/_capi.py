"""ctypes binding of the C ABI in include/insploray_hip.h (libinsploray_hip.so).

This is the Python twin of the `insploray-hip-sys` FFI crate described in INTEGRATION.md: every prototype below
is one declaration of include/insploray_hip.h.  There is no fallback: if the library has not been built
(`make -C path_tracer_amd/csrc` or `__graft_entry__.build()`), importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IR_DEBUG_LIB") or os.path.join(_HERE, "lib", "libinsploray_hip.so")  # IR_DEBUG_LIB: A/B builds in tools/

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP backend first (make -C path_tracer_amd/csrc). "
        "path_tracer_amd has no CPU fallback."
    )



def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own copy (torch/lib/libamdhip64.so + libhsa-runtime64.so)
    and cannot be told to use another; a second copy in the same process finds no GPU (measured: this library on
    /opt/rocm's runtime first, `import torch` afterwards -> "No HIP GPUs are available"; tools/load_order_probe.py).  The
    other order works: the dynamic linker hands torch's already-loaded copy (same SONAME) to this library.  So when torch
    is installed but not imported yet, and no HIP runtime is mapped yet (a profiler's preload may have brought one), load
    torch's copy first.  IR_SYSTEM_HIP_RUNTIME=1 keeps the system runtime (then import torch first, or not at all)."""
    import importlib.util
    import sys
    if os.environ.get("IR_SYSTEM_HIP_RUNTIME") == "1" or "torch" in sys.modules:
        return None
    try:
        with open("/proc/self/maps") as f:
            if any("libamdhip64" in line for line in f):
                return None
        spec = importlib.util.find_spec("torch")
    except (OSError, ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path)
    except OSError:
        return None


_torch_hip_runtime = _share_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)

IR_OK = 0
IR_EINVAL, IR_EDEVICE, IR_ESTATE, IR_ENOMEM, IR_EIO, IR_ELIMIT = -1, -2, -3, -4, -5, -6
IR_SKY_NEAREST, IR_SKY_BILINEAR = 0, 1
IR_PARTITION_NONE, IR_PARTITION_TILES, IR_PARTITION_SAMPLES = 0, 1, 2
IR_ACCEL_AUTO, IR_ACCEL_LINEAR, IR_ACCEL_LBVH = 0, 1, 2
IR_BUILD_AUTO, IR_BUILD_HOST, IR_BUILD_DEVICE, IR_BUILD_REFIT, IR_BUILD_HOST_SAH = 0, 1, 2, 3, 4
IR_SKY_SAMPLING_BSDF, IR_SKY_SAMPLING_MIS = 0, 1
IR_SCHEDULE_AUTO, IR_SCHEDULE_REGENERATION, IR_SCHEDULE_WAVEFRONT = 0, 1, 2
IR_SAMPLE_BLOCK_AUTO = 0
IR_COMM_ID_BYTES = 128

c_float_p = C.POINTER(C.c_float)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)
ctx_p = C.c_void_p

# name -> (restype, argtypes); the single source the symbol-export test iterates over
PROTOTYPES = {
    "ir_abi_version": (C.c_uint32, []),
    "ir_hip_versions": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ir_create": (C.c_int, [C.POINTER(ctx_p), C.c_int]),
    "ir_destroy": (None, [ctx_p]),
    "ir_last_error": (C.c_char_p, [ctx_p]),
    "ir_last_global_error": (C.c_char_p, []),
    "ir_set_stream": (C.c_int, [ctx_p, C.c_void_p]),
    "ir_set_spheres": (C.c_int, [ctx_p, c_float_p, c_i32_p, C.c_uint32]),
    "ir_set_materials": (C.c_int, [ctx_p, c_float_p, C.c_uint32]),
    "ir_set_sky_color": (C.c_int, [ctx_p, c_float_p]),
    "ir_set_skybox": (C.c_int, [ctx_p, c_float_p, C.c_uint32, C.c_uint32]),
    "ir_set_sky_filter": (C.c_int, [ctx_p, C.c_int]),
    "ir_map_output_argb": (C.c_int, [ctx_p, C.POINTER(c_u32_p), C.POINTER(C.c_size_t)]),
    "ir_set_sky_sampling": (C.c_int, [ctx_p, C.c_int]),
    "ir_set_accel": (C.c_int, [ctx_p, C.c_int]),
    "ir_get_accel_info": (C.c_int, [ctx_p, C.POINTER(C.c_int), c_u32_p, c_u32_p]),
    "ir_set_accel_build": (C.c_int, [ctx_p, C.c_int]),
    "ir_get_accel_build_info": (C.c_int, [ctx_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "ir_download_accel": (C.c_int, [ctx_p, C.c_void_p, C.c_uint32, c_float_p, c_i32_p, C.c_uint32]),
    "ir_download_accel_climb": (C.c_int, [ctx_p, C.c_void_p, C.c_uint32, c_i32_p, C.c_uint32]),
    "ir_set_pinhole": (C.c_int, [ctx_p, c_float_p, c_float_p, C.c_float, C.c_float]),
    "ir_set_primary_rays": (C.c_int, [ctx_p, c_float_p, C.c_uint32, C.c_uint32]),
    "ir_pinhole_compute": (C.c_int, [c_float_p, c_float_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                     c_float_p, c_float_p, c_float_p, c_float_p]),
    "ir_pinhole_get_ray": (C.c_int, [c_float_p, c_float_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_uint32, c_float_p]),
    "ir_set_integrator": (C.c_int, [ctx_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "ir_set_schedule": (C.c_int, [ctx_p, C.c_int, C.c_uint32]),
    "ir_set_sample_block": (C.c_int, [ctx_p, C.c_uint32]),
    "ir_get_sample_block": (C.c_int, [ctx_p, C.c_uint32, C.c_uint32, C.c_uint32, c_u32_p]),
    "ir_set_partition": (C.c_int, [ctx_p, C.c_int, C.c_uint32, C.c_uint32]),
    "ir_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ir_comm_init": (C.c_int, [ctx_p, C.c_char_p, C.c_uint32, C.c_uint32]),
    "ir_comm_destroy": (C.c_int, [ctx_p]),
    "ir_combine": (C.c_int, [ctx_p, C.c_uint32]),
    "ir_combine_async": (C.c_int, [ctx_p, C.c_uint32]),
    "ir_combine_wait": (C.c_int, [ctx_p]),
    "ir_last_combine_ms": (C.c_double, [ctx_p]),
    "ir_tile_payload_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ir_pack_owned_tiles": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t]),
    "ir_unpack_tiles": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]),
    "ir_render": (C.c_int, [ctx_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "ir_render_async": (C.c_int, [ctx_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "ir_synchronize": (C.c_int, [ctx_p]),
    "ir_render_noalloc": (C.c_int, [ctx_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "ir_get_output_argb": (C.c_int, [ctx_p, c_u32_p, C.c_size_t]),
    "ir_get_radiance": (C.c_int, [ctx_p, c_float_p, c_u32_p, C.c_size_t]),
    "ir_set_radiance": (C.c_int, [ctx_p, c_float_p, c_u32_p, C.c_size_t]),
    "ir_get_primary_hits": (C.c_int, [ctx_p, c_i32_p, c_float_p, C.c_size_t]),
    "ir_get_size": (C.c_int, [ctx_p, c_u32_p, c_u32_p]),
    "ir_last_render_seconds": (C.c_double, [ctx_p]),
    "ir_last_kernel_ms": (C.c_double, [ctx_p, c_u32_p]),
    "ir_last_paths_kernel_ms": (C.c_double, [ctx_p]),
    "ir_get_paths_kernel_history": (C.c_int, [ctx_p, C.POINTER(C.c_double), C.c_uint32, c_u32_p]),
    "ir_samples_accumulated": (C.c_uint32, [ctx_p]),
    "ir_set_samples_accumulated": (C.c_int, [ctx_p, C.c_uint32]),
    "ir_device_radiance": (C.c_void_p, [ctx_p]),
    "ir_device_counts": (C.c_void_p, [ctx_p]),
    "ir_bind_accumulator": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "ir_enable_stats": (C.c_int, [ctx_p, C.c_int]),
    "ir_get_stats": (C.c_int, [ctx_p, c_u64_p]),
    "ir_get_accel_stats": (C.c_int, [ctx_p, c_u64_p]),
    "ir_get_lane_profile": (C.c_int, [ctx_p, c_u64_p]),
    "ir_get_primary_stats": (C.c_int, [ctx_p, c_u64_p]),
    "ir_selftest_exact_math": (C.c_int, [ctx_p, c_u64_p, C.c_uint32, C.c_uint32]),
    "ir_exr_load": (C.c_int, [C.c_char_p, C.POINTER(c_float_p), c_u32_p, c_u32_p]),
    "ir_exr_last_error": (C.c_char_p, []),
    "ir_exr_save": (C.c_int, [C.c_char_p, c_float_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "ir_png_save": (C.c_int, [C.c_char_p, c_u32_p, C.c_uint32, C.c_uint32]),
    "ir_free": (None, [C.c_void_p]),
    "ir_exr_sample": (C.c_int, [c_float_p, C.c_uint32, C.c_uint32, c_float_p, c_float_p]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class InsplorayError(RuntimeError):
    """Raised where the reference would panic (`unwrap`) or return Err."""

    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def fptr(a):
    """float32 C-contiguous numpy array -> float*"""
    return a.ctypes.data_as(c_float_p)


_runtime_checked = False


def check_hip_runtime():
    """Once per process, after the first context exists: the library was compiled by /opt/rocm's hipcc and may be running
    on the HIP runtime PyTorch bundles (_share_torch_hip_runtime).  That works while the two agree in their major version;
    a mismatch would otherwise surface as obscure registration or launch failures, so say it plainly."""
    global _runtime_checked
    if _runtime_checked:
        return
    _runtime_checked = True
    compiled, runtime = C.c_int(0), C.c_int(0)
    if lib.ir_hip_versions(C.byref(compiled), C.byref(runtime)) != IR_OK or runtime.value == 0:
        return
    if compiled.value // 10000000 != runtime.value // 10000000:
        import warnings
        warnings.warn("libinsploray_hip.so was built against HIP %d.%d but this process runs HIP runtime %d.%d (%s): set "
                      "IR_SYSTEM_HIP_RUNTIME=1 and import torch first, or rebuild against the runtime in use" % (
                          compiled.value // 10000000, compiled.value // 100000 % 100, runtime.value // 10000000,
                          runtime.value // 100000 % 100, "PyTorch's bundled copy" if _torch_hip_runtime is not None else "already loaded"),
                      RuntimeWarning, stacklevel=3)


def check(ctx, rc):
    if rc != IR_OK:
        msg = lib.ir_last_error(ctx) if ctx else lib.ir_last_global_error()
        raise InsplorayError(rc, (msg or b"").decode("utf-8", "replace"))
