"""Headless renderer (the reference's README lists a headless CLI and "save image" as TODOs,
README.md:59-60,104).  Drives the same mirror of the reference API the tests use:

    python -m path_tracer_amd.cli --scene s1 --width 1920 --height 1080 --spp 1024 --out frame.exr
    python -m path_tracer_amd.cli --scene my_scene.json --spp 64 --out frame.ppm --checkpoint acc.npz
    python -m path_tracer_amd.cli --scene my_scene.json --spp 64 --resume acc.npz --out frame.exr   # 64 more samples

Scene JSON (all fields optional except spheres):
    {"default_sky_color": [r,g,b], "skybox": "path.exr" | "synthetic" | null,
     "materials": [{"albedo": [r,g,b], "roughness": x, "metalic": x, "emission_color": [r,g,b], "emissive_power": x}],
     "spheres": [{"position": [x,y,z], "radius": r, "material_id": i}],
     "camera": {"position": [x,y,z], "rotation": [rx,ry,rz], "focal_length": 35, "sensor_size": 55}}

Outputs: .exr = linear mean radiance (sum / count), .png / .ppm = the reference's resolved 0xAARRGGBB frame
(full_buffer.rs:49-65) flipped to top-down as the frontend displays it (app.rs:101-104); the checkpoint is the raw
accumulator (Accumulator.framebuffer / sample_counts) plus the number of samples, so a resumed render continues the
same counter-based sample sequence.  Prints one JSON line of metrics.
"""
import argparse
import json
import sys
import time

import numpy as np

from . import _capi, workloads as W
from .cameras import PinholeCamera
from .file_formats import ExrImage, save_png
from .renderer import RayTracer
from .scene import Matrial, Scene, Sphere


def load_scene(spec, width, height):
    """-> (Scene, PinholeCamera)"""
    if spec == "s1":
        return W.scene_s1(), W.default_camera(width, height)
    if spec == "s1-constant-sky":
        return W.scene_s1(skybox=None), W.default_camera(width, height)
    if spec == "example":
        return Scene.get_example_scene(), W.default_camera(width, height)
    if spec == "s2":
        return W.scene_s2(), W.camera_s2(width, height)
    with open(spec) as f:
        d = json.load(f)
    scene = Scene(default_sky_color=d.get("default_sky_color", (0.0, 0.0, 0.0)))
    for m in d.get("materials", []):
        scene.materials.append(Matrial(albedo=m.get("albedo", (1.0, 1.0, 1.0)), roughness=m.get("roughness", 0.5),
                                       metalic=m.get("metalic", 0.0), emission_color=m.get("emission_color", (0.0, 0.0, 0.0)),
                                       emissive_power=m.get("emissive_power", 0.0)))
    for s in d["spheres"]:
        scene.spheres.append(Sphere(s.get("position", (0.0, 0.0, 0.0)), s.get("radius", 1.0), s.get("material_id", -1)))
    box = d.get("skybox")
    if box == "synthetic":
        scene.skybox = W.synthetic_sky()
    elif box:
        scene.skybox = ExrImage.load_exr_image(box)
    c = d.get("camera", {})
    cam = PinholeCamera(c.get("position", (0.0, 0.0, 2.0)), c.get("rotation", (0.0, 0.0, 0.0)),
                        c.get("focal_length", 35.0), c.get("sensor_size", 55.0), [width, height])
    return scene, cam


def write_ppm(path, argb, width, height):
    """0xAARRGGBB rows bottom-up -> binary PPM top-down."""
    a = np.asarray(argb, dtype=np.uint32).reshape(height, width)[::-1]
    rgb = np.stack([(a >> 16) & 0xFF, (a >> 8) & 0xFF, a & 0xFF], axis=-1).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (width, height))
        f.write(rgb.tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser(prog="path_tracer_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", default="s1", help="s1 | s1-constant-sky | example | s2 | path to a scene JSON")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=450)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=5, help="reference default 5 (renderer.rs:39)")
    ap.add_argument("--rr-start", type=int, default=2, help="max_compulsory_bounces, reference default 2 (renderer.rs:40)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=W.RNG_SEED)
    ap.add_argument("--sky-filter", choices=["nearest", "bilinear"], default="nearest")
    ap.add_argument("--sky-sampling", choices=["bsdf", "mis"], default="bsdf",
                    help="bsdf: the reference's estimator; mis: sky importance sampling (extension)")
    ap.add_argument("--accel", choices=["auto", "linear", "lbvh"], default="auto")
    ap.add_argument("--accel-build", choices=["auto", "host", "device", "sah"], default="auto",
                    help="LBVH builder (ir_set_accel_build); sah: surface-area splits on the host, a slower build and a faster "
                         "tree for a scene rendered with many samples")
    ap.add_argument("--sample-block", default="128",
                    help="samples per accumulation block (ir_set_sample_block): a number, or 'auto' = sized for throughput "
                         "(the order of the sums follows it; 128 is the library default)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", help="image to write: .exr (linear mean radiance), .png or .ppm (resolved frame)")
    ap.add_argument("--checkpoint", help="write the accumulator (.npz) after rendering")
    ap.add_argument("--resume", help="load an accumulator (.npz) written by --checkpoint and add --spp more samples")
    args = ap.parse_args(argv)

    w, h = args.width, args.height
    scene, cam = load_scene(args.scene, w, h)
    rt = RayTracer(0, 0, device=args.device)
    try:
        rt.set_active_camera(cam)
        rt.set_integrator(bounces=args.bounces, max_compulsory_bounces=args.rr_start, seed=args.seed)
        rt.set_sky_filter(_capi.IR_SKY_BILINEAR if args.sky_filter == "bilinear" else _capi.IR_SKY_NEAREST)
        rt.set_sky_sampling(_capi.IR_SKY_SAMPLING_MIS if args.sky_sampling == "mis" else _capi.IR_SKY_SAMPLING_BSDF)
        rt.set_accel({"auto": _capi.IR_ACCEL_AUTO, "linear": _capi.IR_ACCEL_LINEAR, "lbvh": _capi.IR_ACCEL_LBVH}[args.accel])
        rt.set_accel_build({"auto": _capi.IR_BUILD_AUTO, "host": _capi.IR_BUILD_HOST, "device": _capi.IR_BUILD_DEVICE,
                            "sah": _capi.IR_BUILD_HOST_SAH}[args.accel_build])
        rt.set_sample_block(_capi.IR_SAMPLE_BLOCK_AUTO if args.sample_block == "auto" else int(args.sample_block))
        resumed = 0
        if args.resume:
            ck = np.load(args.resume)
            if tuple(ck["size"]) != (w, h):
                raise SystemExit(f"checkpoint is {tuple(ck['size'])}, render asked for {(w, h)}")
            resumed = int(ck["samples"])
            rt.render(scene, w, h, False, spp=0)             # size the film
            rt.set_radiance(ck["sums"], ck["counts"])
            rt.set_samples_accumulated(resumed)              # continue the sample sequence after the stored ones
        t0 = time.perf_counter()
        rt.render(scene, w, h, bool(args.resume), spp=args.spp)
        wall = time.perf_counter() - t0
        kernel_ms, launches = rt.last_kernel_ms()
        sums, counts = rt.get_radiance()
        if args.checkpoint:
            np.savez(args.checkpoint, sums=sums, counts=counts, size=np.array([w, h]), samples=np.array(resumed + args.spp))
        if args.out:
            if args.out.lower().endswith(".png"):
                save_png(args.out, rt.get_output(), w, h)
            elif args.out.lower().endswith(".ppm"):
                write_ppm(args.out, rt.get_output(), w, h)
            else:
                mean = sums[..., :3] / np.maximum(counts, 1)[..., None].astype(np.float32)
                ExrImage(mean[::-1].reshape(-1, 3), w, h).save(args.out)   # EXR rows are top-down
        print(json.dumps({
            "scene": args.scene, "width": w, "height": h, "spp": args.spp, "samples_total": resumed + args.spp,
            "bounces": args.bounces, "rr_start": args.rr_start, "accel": rt.get_accel_info(),
            "sample_block": rt.get_sample_block(w, h, args.spp),
            "render_seconds": wall, "kernel_ms": kernel_ms, "launches": launches,
            "msamples_per_s": w * h * args.spp / max(kernel_ms, 1e-9) / 1e3, "out": args.out,
        }))
    finally:
        rt.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
