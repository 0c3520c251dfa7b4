"""Which HIP / HSA runtime copies end up in a process that uses both this library and PyTorch-ROCm, by import order.
usage: python tools/load_order_probe.py lib-first|torch-first   (IR_SYSTEM_HIP_RUNTIME=1 lib-first shows the failing order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if any(k in p for k in ("libamdhip64", "libhsa-runtime", "librocm", "libhiprtc")):
            seen.add(p)
    return sorted(seen)
if order == "lib-first":
    import path_tracer_amd as pta
    rt = pta.RayTracer(0, 0, device=0)
    print("after the library:", maps())
    import torch
    print("after torch import:", maps())
    try:
        print("torch sees a GPU:", torch.cuda.is_available(), torch.zeros(1, device="cuda").item())
    except Exception as e:
        print("torch failed:", type(e).__name__, str(e)[:100])
else:
    import torch
    print("torch sees a GPU:", torch.cuda.is_available(), torch.zeros(1, device="cuda").item())
    print("after torch:", maps())
    import path_tracer_amd as pta
    rt = pta.RayTracer(0, 0, device=0)
    print("after the library:", maps())
    from path_tracer_amd import workloads as W
    rt.set_active_camera(W.default_camera(64, 36)); rt.render(W.scene_s1(), 64, 36, False, spp=2)
    print("render ok", rt.samples_accumulated())
