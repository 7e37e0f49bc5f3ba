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


def test_tile_roundtrip():
    hdist = importlib.import_module("pbrt-v2_amd.dist")
    f = torch.arange(45 * 70 * 4, dtype=torch.float32).reshape(45, 70, 4)
    t = hdist.film_to_tiles(f)
    assert t.shape == (2 * 3, 32 * 32 * 4)
    assert torch.equal(hdist.tiles_to_film(t, 70, 45), f)
