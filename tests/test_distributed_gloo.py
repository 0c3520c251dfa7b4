"""N > 1 host logic on CPU: world_size 2 (and 3) over gloo.  Each rank produces its partial radiance buffer with the
oracle under the SAME partition semantics the C ABI documents (ir_set_partition), the product's
path_tracer_amd.distributed.combine_partials does the exchange step, and the result is compared with a whole render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from path_tracer_amd import distributed as D, workloads as W
        from tests import oracle_lib as O
        w, h, spp = 150, 100, 4
        scene = W.scene_s1(skybox=None)
        oc = O.camera_from(W.default_camera(w, h), w, h)
        if mode == "tiles":
            s, c, _, _ = O.render(oc, scene, spp, bounces=8, math_mode=1, tile_rank=rank, tile_world=world, threads=2)
            mask = D.owned_pixel_mask(w, h, rank, world).numpy()
            assert ((c == spp) == mask).all() and (c[~mask] == 0).all()   # host tile deal == oracle tile deal
        else:
            base = D.sample_base(rank, world, spp, 0)
            s, c, _, _ = O.render(oc, scene, spp, bounces=8, math_mode=1, sample_base=base, threads=2)
        ts, tc = torch.from_numpy(s), torch.from_numpy(c.astype(np.int32))
        D.combine_partials(ts, tc, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, f"{mode}_sum.npy"), ts.numpy())
            np.save(os.path.join(out_dir, f"{mode}_cnt.npy"), tc.numpy())
        # all-ranks variant leaves the same buffer everywhere
        t2 = torch.full((4,), float(rank + 1))
        c2 = torch.full((4,), rank + 1, dtype=torch.int32)
        D.combine_partials(t2, c2, all_ranks=True)
        assert t2[0].item() == world * (world + 1) / 2 and c2[0].item() == world * (world + 1) // 2
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "tiles"), (2, "samples"), (3, "tiles")])
def test_partition_and_combine(tmp_path, oracle, world, mode):
    from path_tracer_amd import workloads as W
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    s = np.load(tmp_path / f"{mode}_sum.npy")
    c = np.load(tmp_path / f"{mode}_cnt.npy")
    w, h, spp = 150, 100, 4
    scene = W.scene_s1(skybox=None)
    oc = oracle.camera_from(W.default_camera(w, h), w, h)
    if mode == "tiles":
        whole, wc, _, _ = oracle.render(oc, scene, spp, bounces=8, math_mode=1)
        assert np.array_equal(s, whole)            # disjoint supports: x + 0 == x, bit-identical to 1 rank
        assert (c == spp).all()
    else:
        whole, wc, _, _ = oracle.render(oc, scene, spp * world, bounces=8, math_mode=1)
        assert (c == spp * world).all()
        assert np.allclose(s, whole, rtol=1e-5, atol=1e-5)   # same samples, different f32 summation order


def test_helpers():
    from path_tracer_amd import distributed as D
    assert D.tile_grid(1920, 1080) == (30, 17) and D.tile_grid(800, 450) == (13, 8) and D.tile_grid(3840, 2160) == (60, 34)
    m = [D.owned_pixel_mask(130, 70, r, 3) for r in range(3)]
    assert (sum(x.int() for x in m) == 1).all()
    assert D.sample_base(1, 4, 16, 32) == 32 * 4 + 16
    # single process: combine is the identity
    a, b = torch.ones(3), torch.ones(3, dtype=torch.int32)
    assert D.combine_partials(a, b)[0] is a
