"""Headless CLI on the GPU: JSON scene, image outputs, checkpoint/resume continues the same sample sequence."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cli_checkpoint_resume_bit_identical(tmp_path, capsys):
    from path_tracer_amd import cli
    from path_tracer_amd.file_formats import ExrImage
    scene = {
        "default_sky_color": [0.6, 0.7, 0.9], "skybox": "synthetic",
        "materials": [{"albedo": [1, 0, 1]}, {"albedo": [0.2, 0.3, 1.0]},
                      {"albedo": [1, 1, 1], "emission_color": [1.0, 0.8, 0.6], "emissive_power": 1.0}],
        "spheres": [{"position": [0, 0, 0], "radius": 0.5, "material_id": 0},
                    {"position": [0, -100.5, 0], "radius": 100.0, "material_id": 1},
                    {"position": [1.0, 0.25, -0.5], "radius": 0.25, "material_id": 2}],
        "camera": {"position": [0, 0, 2], "rotation": [0, 0, 0], "focal_length": 35, "sensor_size": 55},
    }
    sp = tmp_path / "scene.json"
    sp.write_text(json.dumps(scene))
    common = ["--scene", str(sp), "--width", "96", "--height", "54", "--bounces", "8"]
    assert cli.main(common + ["--spp", "8", "--checkpoint", str(tmp_path / "full.npz"), "--out", str(tmp_path / "f.exr")]) == 0
    assert cli.main(common + ["--spp", "5", "--checkpoint", str(tmp_path / "half.npz")]) == 0
    assert cli.main(common + ["--spp", "3", "--resume", str(tmp_path / "half.npz"), "--checkpoint", str(tmp_path / "resumed.npz"),
                              "--out", str(tmp_path / "f.ppm")]) == 0
    lines = [json.loads(x) for x in capsys.readouterr().out.strip().splitlines()]
    assert lines[-1]["samples_total"] == 8 and lines[0]["accel"]["nodes"] == 0
    a, b = np.load(tmp_path / "full.npz"), np.load(tmp_path / "resumed.npz")
    # resumed == the same two calls made in one process (bitwise): samples 0-4, then samples 5-7 merged on top
    import path_tracer_amd as pta
    from path_tracer_amd import workloads as W
    rt = pta.RayTracer(0, 0, device=0)
    try:
        rt.set_active_camera(W.default_camera(96, 54))
        rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
        rt.render(W.scene_s1(), 96, 54, False, spp=5)
        rt.render(W.scene_s1(), 96, 54, True, spp=3)
        two_calls, cnt = rt.get_radiance()
    finally:
        rt.close()
    assert np.array_equal(b["sums"].view(np.uint32), two_calls.view(np.uint32)) and np.array_equal(b["counts"], cnt)
    # --sample-block auto: the metrics line says which block size the render used (a small frame at 64 spp: 4)
    assert cli.main(common + ["--spp", "64", "--sample-block", "auto"]) == 0
    assert cli.main(common + ["--spp", "64"]) == 0
    auto, default = [json.loads(x) for x in capsys.readouterr().out.strip().splitlines()][-2:]
    assert auto["sample_block"] == 4 and default["sample_block"] == 128
    # ... and the same 8 samples as the single 8-sample call, up to the order of the f32 additions
    assert np.array_equal(a["counts"], b["counts"]) and np.allclose(a["sums"], b["sums"], rtol=1e-5, atol=1e-6)
    assert not np.array_equal(a["sums"][..., :3], np.load(tmp_path / "half.npz")["sums"][..., :3])
    # the JSON scene is S1: same buffer as the built-in workload
    assert cli.main(["--scene", "s1", "--width", "96", "--height", "54", "--bounces", "8", "--spp", "8",
                     "--checkpoint", str(tmp_path / "s1.npz")]) == 0
    assert np.array_equal(np.load(tmp_path / "s1.npz")["sums"], a["sums"])
    img = ExrImage.load_exr_image(str(tmp_path / "f.exr"))
    mean = a["sums"][..., :3] / 8.0
    assert (img.width, img.height) == (96, 54) and np.allclose(img.pixels_buffer.reshape(54, 96, 3), mean[::-1], rtol=1e-6)
    ppm = (tmp_path / "f.ppm").read_bytes()
    assert ppm.startswith(b"P6\n96 54\n255\n") and len(ppm) == len(b"P6\n96 54\n255\n") + 96 * 54 * 3


def test_cli_surface_area_build_same_bits(tmp_path, capsys):
    """--accel-build sah: another tree over the same 10 000 spheres, the same accumulator bit for bit."""
    from path_tracer_amd import cli
    common = ["--scene", "s2", "--width", "64", "--height", "36", "--bounces", "6", "--spp", "4", "--accel", "lbvh"]
    assert cli.main(common + ["--checkpoint", str(tmp_path / "auto.npz")]) == 0
    assert cli.main(common + ["--accel-build", "sah", "--checkpoint", str(tmp_path / "sah.npz")]) == 0
    lines = [json.loads(x) for x in capsys.readouterr().out.strip().splitlines()]
    assert lines[0]["accel"]["nodes"] > 1000 and lines[1]["accel"]["nodes"] > 1000
    assert lines[0]["accel"]["nodes"] != lines[1]["accel"]["nodes"]  # really another tree
    a, b = np.load(tmp_path / "auto.npz"), np.load(tmp_path / "sah.npz")
    assert a["sums"].tobytes() == b["sums"].tobytes() and a["counts"].tobytes() == b["counts"].tobytes()
