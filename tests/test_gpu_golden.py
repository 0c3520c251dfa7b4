"""HIP path vs the committed golden vectors (no oracle at run time)."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_r1.npz"))


@pytest.fixture(scope="module")
def tracer():
    import path_tracer_amd as pta
    rt = pta.RayTracer(0, 0, device=0)
    yield rt
    rt.close()


def test_primary_hits_golden(tracer):
    from path_tracer_amd import workloads as W
    tracer.set_active_camera(W.default_camera(800, 450))
    idx, t = tracer.get_primary_hits(W.scene_s1(), 800, 450, with_t=True)
    assert [zlib.crc32(idx.tobytes()), zlib.crc32(t.tobytes())] == list(G["s1_hits_800x450_crc"])
    tracer.set_active_camera(W.camera_s2(480, 270))
    idx, t = tracer.get_primary_hits(W.scene_s2(), 480, 270, with_t=True)
    assert [zlib.crc32(idx.tobytes()), zlib.crc32(t.tobytes())] == list(G["s2_hits_480x270_crc"])
    assert np.array_equal(idx[100:136, 200:264], G["s2_hits_crop"].astype(np.int32))


def test_radiance_golden(tracer):
    from path_tracer_amd import workloads as W
    scene = W.scene_s1()
    tracer.set_active_camera(W.default_camera(64, 36))
    tracer.set_integrator(bounces=8, max_compulsory_bounces=2, seed=W.RNG_SEED)
    tracer.enable_stats()
    tracer.render(scene, 64, 36, False, spp=8)
    s, c = tracer.get_radiance()
    st = tracer.get_stats()
    assert np.array_equal(s.view(np.uint32), G["s1_sum_64x36_8spp_math1"].view(np.uint32))   # bit-exact
    assert [st[k] for k in ("samples", "traces", "tests", "disc_pos", "hits")] == list(G["s1_stats_math1"])
    g0 = G["s1_sum_64x36_8spp_math0"].astype(np.float64)[..., :3]                              # libm oracle
    d = s.astype(np.float64)[..., :3] - g0
    assert np.sqrt((d ** 2).sum(-1).mean()) / np.sqrt((g0 ** 2).sum(-1).mean()) < 1e-3
    tracer.set_integrator(bounces=5)
    tracer.render(W.scene_s1(skybox=None), 64, 36, False, spp=8)
    s, _ = tracer.get_radiance()
    assert np.array_equal(s.view(np.uint32), G["s1_nosky_sum_64x36_8spp_b5_math1"].view(np.uint32))


def test_resolve_golden(tracer):
    from path_tracer_amd import workloads as W
    sums, counts = G["resolve_sums"], G["resolve_counts"]
    tracer.set_active_camera(W.default_camera(6, 1))
    tracer.render(W.scene_s1(skybox=None), 6, 1, False, spp=0)
    tracer.set_radiance(sums.reshape(1, 6, 4), counts.reshape(1, 6))
    out = tracer.get_output()
    ref = G["resolve_argb"]
    for shift in (0, 8, 16, 24):
        assert np.abs(((out >> shift) & 0xFF).astype(int) - ((ref >> shift) & 0xFF).astype(int)).max() <= 1
    assert out[0] == 0
