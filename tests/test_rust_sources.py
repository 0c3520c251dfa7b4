"""The Rust host layer (rust/) cannot be compiled here (no cargo / rustc in the image), so these tests keep its sources
consistent with what IS built and tested: the Makefile's source list and flags, the C header through the ctypes prototype
table (tests/test_capi_symbols.py checks that table against the header and the library), and the façade's own calls.
They are text checks; they do not claim that the crate compiles."""
import ctypes
import re
from pathlib import Path

from path_tracer_amd import _capi

ROOT = Path(__file__).resolve().parents[1]
SYS_LIB = ROOT / "rust/insploray-hip-sys/src/lib.rs"
BUILD_RS = ROOT / "rust/insploray-hip-sys/build.rs"
FACADE = ROOT / "rust/insploray/src/lib.rs"
MAKEFILE = ROOT / "path_tracer_amd/csrc/Makefile"


def _make_var(name):
    m = re.search(rf"^{name}\s*=\s*(.*)$", MAKEFILE.read_text(), re.M)
    assert m, name
    return m.group(1).split()


def _rust_list(name):
    m = re.search(rf"const {name}: &\[&str\] = &\[(.*?)\];", BUILD_RS.read_text(), re.S)
    assert m, name
    return re.findall(r'"([^"]+)"', m.group(1))


def _externs():
    """name -> (argument type strings, return type string or None) of every function in the extern "C" block"""
    text = SYS_LIB.read_text()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, re.S):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (args, m.group(3))
    return out


def test_build_rs_compiles_what_the_makefile_compiles():
    assert _rust_list("SRCS") == _make_var("SRCS")
    cxx = [f for f in _make_var("CXXFLAGS") if not f.startswith("-W")]
    assert _rust_list("FLAGS") == cxx
    link_line = next(l for l in MAKEFILE.read_text().splitlines() if "-shared" in l)
    assert _rust_list("LIBS") == re.findall(r"-l\w+", link_line)
    arch = re.search(r"^ARCH\s*\??=\s*(\S+)", MAKEFILE.read_text(), re.M).group(1)
    assert f"--offload-arch={arch}" in BUILD_RS.read_text()
    headers = {Path(h).name for h in _make_var("HDRS")} - {"insploray_hip.h"}
    assert headers <= set(_rust_list("HDRS"))
    for f in _rust_list("SRCS") + _rust_list("HDRS"):
        assert (MAKEFILE.parent / f).exists(), f


def test_every_extern_is_in_the_tested_prototype_table_and_back():
    ext = _externs()
    assert set(ext) == set(_capi.PROTOTYPES), set(ext) ^ set(_capi.PROTOTYPES)


_SCALARS = {
    "c_int": ctypes.c_int, "i32": ctypes.c_int, "u32": ctypes.c_uint, "u64": ctypes.c_uint64, "usize": ctypes.c_size_t,
    "c_float": ctypes.c_float, "f32": ctypes.c_float, "c_double": ctypes.c_double, "f64": ctypes.c_double,
}


def _same_type(rust, ct):
    if rust is None:
        return ct is None
    rust = rust.strip()
    if rust in _SCALARS:
        return ct is _SCALARS[rust] or ctypes.sizeof(ct) == ctypes.sizeof(_SCALARS[rust]) and ct.__name__.lstrip("c_u") == _SCALARS[rust].__name__.lstrip("c_u")
    # every pointer is a pointer: ctypes side is c_void_p, c_char_p or POINTER(...)
    return rust.startswith("*") and (ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, "contents"))


def test_extern_signatures_match_the_prototype_table():
    for name, (args, ret) in _externs().items():
        res, argtypes = _capi.PROTOTYPES[name]
        assert len(args) == len(argtypes), name
        for i, (r, c) in enumerate(zip(args, argtypes)):
            assert _same_type(r, c), (name, i, r, c)
        assert _same_type(ret, res), (name, ret, res)


def test_facade_only_calls_declared_functions_and_constants():
    sys_text = SYS_LIB.read_text()
    declared = set(_externs()) | set(re.findall(r"pub const (\w+)", sys_text)) | set(re.findall(r"pub struct (\w+)", sys_text))
    used = set(re.findall(r"\bsys::(\w+)", FACADE.read_text()))
    assert used and used <= declared, used - declared
    # the calls the hot path needs (renderer.rs:95-169) are really made, not described
    for must in ("ir_create", "ir_set_spheres", "ir_set_materials", "ir_set_sky_color", "ir_set_skybox", "ir_set_pinhole",
                 "ir_set_primary_rays", "ir_set_integrator", "ir_render", "ir_map_output_argb", "ir_destroy",
                 "ir_pinhole_compute", "ir_pinhole_get_ray", "ir_exr_load", "ir_exr_sample", "ir_free", "ir_combine"):
        assert must in used, must


def test_facade_has_the_reference_surface():
    text = FACADE.read_text()
    for item in ("pub mod cameras", "pub mod file_formats", "pub mod renderer", "pub mod scene", "pub use glam::{Vec2, Vec3}",
                 "pub trait Camera", "pub type SharedCamera", "pub struct PinholeCamera", "pub struct RayTracer",
                 "pub struct Scene", "pub struct Sphere", "pub struct Matrial", "pub struct ExrImage",
                 "pub fn new(width: u32, height: u32) -> Self", "pub fn set_active_camera(&mut self, camera: SharedCamera)",
                 "pub fn get_current_size(&self) -> [u32; 2]",
                 "pub fn prepare_pixels(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32)",
                 "pub fn render_updated(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32)",
                 "pub fn render(&mut self, scene: &Arc<RwLock<Scene>>, width: u32, height: u32, acc: bool)",
                 "pub fn get_output(&mut self) -> &[u32]", "pub fn get_last_render_time(&self) -> Duration",
                 "pub fn get_example_scene() -> Self", "pub fn sample(&self, dir: Vec3) -> Vec3",
                 "pub fn load_exr_image(path: &str)", "pub fn get_camera_to_world_matrix(&self) -> Mat4",
                 "pub fn set_focal_length(&mut self, focal_length: f32)", "pub fn set_sensor_size(&mut self, sensor_size: f32)"):
        assert item in text, item
    # no method body left as a comment: every `fn` has statements, and delimiters balance
    code = re.sub(r"//[^\n]*", "", text)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)
    for o, c in ("{}", "()", "[]"):
        assert code.count(o) == code.count(c), (o, code.count(o), code.count(c))
    for m in re.finditer(r"fn (\w+)[^{;]*\{\s*\}", code):
        raise AssertionError(f"empty body: {m.group(1)}")
