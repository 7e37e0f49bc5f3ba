"""N>1 path on CPU: world_size-2 gloo processes, each producing its shard's film with the
test-only host emulation of the device code, gathered with pbrt-v2_amd/dist.py; the result must
equal the single-rank film."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import ROOT, abi, hash_rd, load_case


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.hostemu import emu
    hdist = importlib.import_module("pbrt-v2_amd.dist")
    s = load_case("k8")
    rd = hash_rd(s, seed=4, spp=2)
    rd.shard_rank, rd.shard_count = rank, world
    f, _ = emu.EmuScene(s).render(s.camera, rd)
    full = hdist.gather_film(torch.from_numpy(f), rank, world)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_and_gather_equals_single_rank(tmp_path, world):
    from tests.hostemu import emu
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    s = load_case("k8")
    rd = hash_rd(s, seed=4, spp=2)
    want, _ = emu.EmuScene(s).render(s.camera, rd)
    # own-pixel sums are exact; the rare cross-tile spills may be summed in another order
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(got[..., 3], want[..., 3])


def _worker_filtered(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.hostemu import emu
    hdist = importlib.import_module("pbrt-v2_amd.dist")
    s = load_case("fmitch")
    rd = hash_rd(s, seed=4)
    rd.shard_rank, rd.shard_count = rank, world
    f, _ = emu.EmuScene(s).render(s.camera, rd, flt=s.filter)
    full = hdist.exchange_film(torch.from_numpy(f), rank, world, wide_filter=True)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_render_under_a_wide_filter_reduces_to_the_single_rank_film(tmp_path):
    """with PixelFilter "mitchell" 3 x 2.5 the shards overlap at tile borders; the exchange is a
    sum-reduce of full-frame partial films."""
    from tests.hostemu import emu
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker_filtered, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    s = load_case("fmitch")
    rd = hash_rd(s, seed=4)
    want, _ = emu.EmuScene(s).render(s.camera, rd, flt=s.filter)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_tile_records_with_aprons_carry_cross_shard_boundary_samples_exactly_once(world):
    """Round 4 (found on the GPU by the two-process host-transport test): a camera sample on an exact pixel boundary lands in two pixels
    (film/image.cpp:82-89); when the second pixel lies in a tile of another shard, a gather of bare 32 x 32 tiles dropped it.  Tile records
    carry a one-pixel apron; each spilled pixel is carried by exactly ONE record of the spilling shard (apron_carrier).  Synthetic shard
    films: own tiles filled, plus a spill into EVERY pixel adjacent (8-neighbourhood) to an own tile; the sum of the records must be the
    sum of the films, pixel for pixel — no spill lost, none counted twice."""
    hdist = importlib.import_module("pbrt-v2_amd.dist")
    H, W = 100, 150          # 5 x 4 tiles, ragged at both edges
    nx, ny = hdist.tile_grid(W, H)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:H, 0:W]
    owner = ((yy >> 5) * nx + (xx >> 5)) % world
    total = np.zeros((H, W, 4), dtype=np.float64)
    films = []
    for r in range(world):
        own = owner == r
        near = np.zeros_like(own)                     # pixels within one pixel of an own pixel
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                sh = np.zeros_like(own)
                ys, yd = slice(max(dy, 0), H + min(dy, 0)), slice(max(-dy, 0), H + min(-dy, 0))
                xs, xd = slice(max(dx, 0), W + min(dx, 0)), slice(max(-dx, 0), W + min(-dx, 0))
                sh[yd, xd] = own[ys, xs]
                near |= sh
        f = np.zeros((H, W, 4), dtype=np.float32)
        vals = rng.integers(1, 9, size=(H, W, 4)).astype(np.float32)      # small integers: float sums are exact in any order
        f[near] = vals[near]
        films.append(f)
        total += f
    out = films[0].copy()
    for r in range(1, world):
        recs = hdist.film_to_records(films[r], r, world)
        assert recs.shape == (len(range(r, nx * ny, world)), hdist.TW * hdist.TW * 4)
        hdist.add_records(out, np.asarray(recs), r, world)
    assert np.array_equal(out.astype(np.float64), total)
