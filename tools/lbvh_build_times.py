"""Host vs device LBVH build time per sphere count (ir_get_accel_build_info: wall time of the build incl. uploads and
synchronisation).  Run on the GPU box: python tools/lbvh_build_times.py [sphere counts] > gpurun_out/lbvh_build_times.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402


def main():
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_accel(_capi.IR_ACCEL_LBVH)
    print(f"{'spheres':>9} {'host ms':>10} {'device ms':>10} {'ratio':>7} {'refit ms':>9} {'nodes':>8} {'depth':>5}")
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 10000, 30000, 100000, 300000, 1000000, 4000000]
    for n in sizes:
        base = W.scene_s2(n=n)
        res = {}
        for name, mode in (("host", _capi.IR_BUILD_HOST), ("device", _capi.IR_BUILD_DEVICE), ("refit", _capi.IR_BUILD_REFIT)):
            rt.set_accel_build(mode)
            times = []
            reps = 7 if n <= 300000 else 3
            for k in range(reps):
                xyzr = base.xyzr.copy()
                xyzr[1:, 0] += np.float32(0.01 * (k + 1))  # an edit, so that the tree is rebuilt
                scene = W.FlatScene(xyzr, base.material_id, base.materials9, default_sky_color=(0.6, 0.7, 0.9), skybox=None)
                rt._upload_scene(scene)
                info = rt.get_accel_build_info()
                assert info["builder"] == mode or (name == "refit" and k == 0), info
                times.append(info["seconds"] * 1e3)
            res[name] = float(np.median(times[1:]))
            acc = rt.get_accel_info()
        print(f"{n:>9} {res['host']:>10.3f} {res['device']:>10.3f} {res['host'] / res['device']:>7.2f} {res['refit']:>9.3f} {acc['nodes']:>8} {acc['depth']:>5}", flush=True)
    rt.close()


if __name__ == "__main__":
    main()
