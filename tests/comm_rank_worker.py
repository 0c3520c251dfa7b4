"""One rank of tests/test_gpu_comm_ranks.py: a fresh process that renders its share of a frame on device 0, joins the
communicator (through the stand-in transport named by IR_DEBUG_RCCL_LIB) and takes part in the exchange step.
usage: comm_rank_worker.py <rank> <world> <tiles|samples> <id file> <out .npz (root only)> <w> <h> <spp>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import path_tracer_amd as pta  # noqa: E402
from path_tracer_amd import _capi, workloads as W  # noqa: E402
from path_tracer_amd.distributed import NativeExchange  # noqa: E402

rank, world, mode, id_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
w, h, spp = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
ROOT_RANK = int(os.environ.get("IR_TEST_ROOT", "0"))


def rendezvous(ident):
    """rank 0 writes the id (atomically), the others wait for the file"""
    if rank == 0:
        with open(id_file + ".tmp", "wb") as f:
            f.write(ident)
        os.replace(id_file + ".tmp", id_file)
        return ident
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            raise SystemExit("rank %d: no rendezvous id after 60 s" % rank)
        time.sleep(0.01)
    with open(id_file, "rb") as f:
        return f.read()


rt = pta.RayTracer(0, 0, device=0)
rt.set_active_camera(W.default_camera(w, h))
rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
rt.set_partition(_capi.IR_PARTITION_TILES if mode == "tiles" else _capi.IR_PARTITION_SAMPLES, rank, world)
scene = W.scene_s1()
ex = NativeExchange(rt, rank, world, broadcast_id=rendezvous)
out = {}
# 1: blocking combine of one render
rt.render(scene, w, h, False, spp=spp)
ex.combine(ROOT_RANK)
if rank == ROOT_RANK:
    s, c = rt.get_radiance()
    out["s1"], out["c1"] = s.copy(), c.copy()
# 2: accumulate on, then the overlapped form: snapshot + exchange on its own stream, the next render enqueued right behind
rt.render(scene, w, h, True, spp=spp)
ex.combine_async(ROOT_RANK)
rt.render(scene, w, h, True, spp=1, blocking=False)
ex.wait()
rt.synchronize()
if rank == ROOT_RANK:
    # readbacks after a render show the rank's own film again; the combined film of exchange 2 is what a combine shows
    pass
# 3: a third exchange (buffers and counters of the transport reused), of the film with 2 spp + 1 samples
ex.combine(ROOT_RANK)
if rank == ROOT_RANK:
    s, c = rt.get_radiance()
    out["s3"], out["c3"] = s.copy(), c.copy()
    out["ms"] = np.array([ex.last_ms()])
    np.savez(out_file, **out)
ex.close()
rt.close()
print("rank %d of %d done" % (rank, world), flush=True)
