#!/usr/bin/env python3
"""Empirical cost split of the render kernel on C2 (256 spp): variants that remove one section each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import path_tracer_amd as pta
from path_tracer_amd import workloads as W

w, h, spp = 1920, 1080, 256
rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h))
def run(name, scene, bounces, rr):
    rt.set_integrator(bounces=bounces, max_compulsory_bounces=rr, seed=W.RNG_SEED)
    best = 1e9
    for _ in range(3):
        rt.render(scene, w, h, False, spp=spp)
        best = min(best, rt.last_kernel_ms()[0])
    rt.enable_stats(); rt.render(scene, w, h, False, spp=spp); st = rt.get_stats()
    print(f"{name:44s} {best:7.3f} ms  traces/sample {st['traces']/st['samples']:.3f} hits/sample {st['hits']/st['samples']:.3f}", flush=True)
sky = W.scene_s1()
nosky = W.scene_s1(skybox=None)
run("C2 (B=8, R=2, skybox)", sky, 8, 2)
run("constant sky", nosky, 8, 2)
run("B=1 (shade only, no bounce trace)", sky, 1, 2)
run("B=2", sky, 2, 2)
run("B=2 constant sky", nosky, 2, 2)
run("B=8 no roulette (R=8)", sky, 8, 8)
run("B=8 roulette from 0", sky, 8, 0)
run("B=0", sky, 0, 2)
rt.close()
