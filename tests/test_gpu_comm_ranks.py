"""The N-rank branches of the exchange step (csrc/insploray_comm.hip: the root's grouped ncclRecv per peer, payload
offsets per rank, the unpack loop, the grouped two-datatype ncclReduce), executed with 2 and 3 ranks on the one GPU a
test box has.  RCCL refuses two ranks on one device, so the ranks — fresh processes, tests/comm_rank_worker.py — run the
library against a stand-in transport with RCCL's entry points (tests/cpp/fake_rccl.cpp, IR_DEBUG_RCCL_LIB: shared memory
+ host staging).  What this pins: ir_comm_init / ir_combine / ir_combine_async / ir_combine_wait hand a conforming
transport the right buffers, counts, peers and offsets, and the root ends up with the film one GPU renders — bit for bit
for the tile partition, within f32 summation order for the sample-range partition.  What it does not: RCCL and xGMI
themselves (unmeasured on hardware until a SCALE record exists).
Reference analogue: the merge loop renderer.rs:152-159."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
WORKER = os.path.join(ROOT, "tests", "comm_rank_worker.py")


@pytest.fixture(scope="module")
def tracer():
    import path_tracer_amd as pta
    from path_tracer_amd import workloads as W
    rt = pta.RayTracer(0, 0, device=0)
    rt.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
    yield rt
    rt.close()


def _run_ranks(tmp_path, world, mode, w, h, spp, root=0):
    if not os.path.exists(FAKE):
        pytest.fail(f"{FAKE} is missing: __graft_entry__.build() compiles it")
    env = dict(os.environ, IR_DEBUG_RCCL_LIB=FAKE, IR_TEST_ROOT=str(root))
    env.pop("IR_DEBUG_LIB", None)
    id_file, out_file = str(tmp_path / "id.bin"), str(tmp_path / "root.npz")
    procs = [subprocess.Popen([sys.executable, "-u", WORKER, str(r), str(world), mode, id_file, out_file, str(w), str(h), str(spp)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank did not finish in 150 s")
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-2000:]}"
    return np.load(out_file)


@pytest.mark.parametrize("world,root", [(2, 0), (3, 0), (3, 2)])
def test_tile_gather_across_ranks_is_the_one_gpu_film(tracer, tmp_path, world, root):
    from path_tracer_amd import workloads as W
    w, h, spp = 330, 200, 6      # 6 x 4 tiles, ragged edges; 24 tiles do not divide by... 3 ranks own 8 each, 2 own 12
    got = _run_ranks(tmp_path, world, "tiles", w, h, spp, root)
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(w, h))
    tracer.render(scene, w, h, False, spp=spp)
    s1, c1 = tracer.get_radiance()
    assert got["c1"].tobytes() == c1.tobytes() and got["s1"].tobytes() == s1.tobytes()
    tracer.render(scene, w, h, True, spp=spp)
    tracer.render(scene, w, h, True, spp=1)
    s3, c3 = tracer.get_radiance()
    assert (c3 == 2 * spp + 1).all()
    assert got["c3"].tobytes() == c3.tobytes() and got["s3"].tobytes() == s3.tobytes()
    assert float(got["ms"][0]) > 0.0


@pytest.mark.parametrize("world,root", [(2, 0), (3, 1)])
def test_sample_range_reduce_across_ranks(tracer, tmp_path, world, root):
    """Every rank renders its own sample indices of every pixel; the root's film is their sum in rank order (the stand-in
    adds in rank order; RCCL's order is its own: 1e-6 relative is the bar either way)."""
    from path_tracer_amd import _capi, workloads as W
    w, h, spp = 200, 120, 5
    got = _run_ranks(tmp_path, world, "samples", w, h, spp, root)
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(w, h))
    want = np.zeros((h, w, 4), dtype=np.float32)
    try:
        for r in range(world):      # one GPU plays every rank in turn; summed here in rank order too
            tracer.set_partition(_capi.IR_PARTITION_SAMPLES, r, world)
            tracer.render(scene, w, h, False, spp=spp)
            s, c = tracer.get_radiance()
            assert (c == spp).all()
            want = want + s if r else s.copy()
    finally:
        tracer.set_partition(_capi.IR_PARTITION_NONE, 0, 1)
    assert (got["c1"] == spp * world).all()
    assert got["s1"].tobytes() == want.tobytes()          # same order of additions: the same bits
    # against the unpartitioned render of the same sample indices: f32 summation order only
    tracer.render(scene, w, h, False, spp=spp * world)
    whole, _ = tracer.get_radiance()
    ok = np.isfinite(whole).all(-1)
    assert np.allclose(got["s1"][ok], whole[ok], rtol=1e-5, atol=1e-5)
    assert (got["c3"] == (2 * spp + 1) * world).all()
