"""Convergence of the two sky estimators on workload C5 (S1 without emitter, sun-dominated sky): relative RMSE against a
converged image per sample count, throughput, and the time each needs for the same error.
Run on the GPU box: python tools/sky_sampling_convergence.py > gpurun_out/sky_sampling.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402


def main():
    w, h = 1920, 1080
    scene = W.scene_s1(emitter=False, skybox=W.synthetic_sky(sun_scale=4000.0))
    cam = W.default_camera(w, h)
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_active_camera(cam)

    def render(mode, spp, seed):
        rt.set_sky_sampling(mode)
        rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=seed)
        rt.render(scene, w, h, False, spp=spp)
        ms = rt.last_kernel_ms()[0]
        s, c = rt.get_radiance()
        return s[..., :3].astype(np.float64) / c[..., None], ms

    # converged image: both estimators, independent seeds, weighted by inverse variance would be best; MIS alone at
    # 32768 spp has an error far below every row of the table
    ref, _ = render(_capi.IR_SKY_SAMPLING_MIS, 32768, 0xABCDEF)
    chk, _ = render(_capi.IR_SKY_SAMPLING_BSDF, 32768, 0x13579B)
    # The reference's weight brdf cos / pdf is 0/0 when the sampled cosine rounds to zero (about once per 2^24 bounces);
    # like the reference, both estimators then carry a NaN in that pixel's sum.  Such pixels are left out of the errors.
    ok = np.isfinite(ref).all(-1) & np.isfinite(chk).all(-1)

    def err(img):
        m = ok & np.isfinite(img).all(-1)
        return np.sqrt(((img[m] - ref[m]) ** 2).sum(-1).mean()) / np.sqrt((ref[m] ** 2).sum(-1).mean())

    print(f"C5 {w}x{h}, depth 8; reference = MIS at 32768 spp; BSDF at 32768 spp differs from it by "
          f"rel-RMSE {err(chk):.4f}, mean ratio {chk[ok].mean() / ref[ok].mean():.5f}; "
          f"{int((~ok).sum())} of {w * h} pixels hold a NaN (0/0 weight, as in the reference) and are left out")
    print(f"{'spp':>6} | {'BSDF rel-RMSE':>13} {'ms':>8} {'Gsamples/s':>10} | {'MIS rel-RMSE':>12} {'ms':>8} {'Gsamples/s':>10} | "
          f"{'error ratio':>11} {'equal-error speed-up':>20}")
    for spp in (16, 64, 256, 1024, 4096):
        row = {}
        for mode in (_capi.IR_SKY_SAMPLING_BSDF, _capi.IR_SKY_SAMPLING_MIS):
            img, ms = render(mode, spp, 0x0123456789ABCDEF)
            row[mode] = (err(img), ms)
        (eb, tb), (em, tm) = row[_capi.IR_SKY_SAMPLING_BSDF], row[_capi.IR_SKY_SAMPLING_MIS]
        # error ~ 1/sqrt(samples): time for BSDF to reach MIS's error = tb * (eb/em)^2
        print(f"{spp:>6} | {eb:>13.4f} {tb:>8.2f} {w * h * spp / tb / 1e6:>10.1f} | {em:>12.4f} {tm:>8.2f} "
              f"{w * h * spp / tm / 1e6:>10.1f} | {eb / em:>11.1f} {tb * (eb / em) ** 2 / tm:>20.0f}", flush=True)
    rt.close()


if __name__ == "__main__":
    main()
