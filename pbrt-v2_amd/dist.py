"""Multi-GPU: pixel-tile sharding + the film gather.

The path shards with NO data-path collective: the scene is replicated in each GPU's HBM, rank r
renders the 32x32 super-tiles t with t % world == r (hpt_render_desc.shard_rank/shard_count —
the same round-robin the kernel's work counter walks), and the only communication is ONE gather
of film tiles to rank 0 at end of frame: 16 B/pixel, 33 MB at 1080p, point-to-point over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

Under a reconstruction filter wider than the default box (hpt_scene_set_filter) a sample reaches
pixels of neighbouring tiles, so every rank's film holds partial sums over the whole frame and the exchange is one
sum-reduce of the full-frame films to rank 0 (reduce_film; 33 MB at 1080p — ring-reduced over xGMI it is per-link
bound and still well under a millisecond).
"""
import numpy as np
import torch
import torch.distributed as dist

TILE = 32


def tile_grid(x_count, y_count):
    return (x_count + TILE - 1) // TILE, (y_count + TILE - 1) // TILE


def film_to_tiles(film):
    """film (H, W, 4) -> (n_tiles, TILE*TILE*4), zero-padded at the right/bottom edge."""
    H, W, _ = film.shape
    nx, ny = tile_grid(W, H)
    pad = torch.zeros((ny * TILE, nx * TILE, 4), dtype=film.dtype, device=film.device)
    pad[:H, :W] = film
    return pad.view(ny, TILE, nx, TILE, 4).permute(0, 2, 1, 3, 4).reshape(ny * nx, TILE * TILE * 4)


def tiles_to_film(tiles, x_count, y_count):
    nx, ny = tile_grid(x_count, y_count)
    f = tiles.view(ny, nx, TILE, TILE, 4).permute(0, 2, 1, 3, 4).reshape(ny * TILE, nx * TILE, 4)
    return f[:y_count, :x_count].contiguous()


def gather_film(film, rank, world, group=None):
    """Each rank passes its full-frame film tensor (only its own tiles are non-zero); rank 0
    receives the assembled frame, other ranks get None.  One collective, tiles only."""
    if world == 1:
        return film
    H, W, _ = film.shape
    tiles = film_to_tiles(film)
    nt = tiles.shape[0]
    per = (nt + world - 1) // world
    mine = tiles[rank::world]
    send = torch.zeros((per, tiles.shape[1]), dtype=film.dtype, device=film.device)
    send[:mine.shape[0]] = mine
    if rank == 0:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=0, group=group)
        out = torch.zeros_like(tiles)
        for r in range(world):
            n = out[r::world].shape[0]
            out[r::world] = recv[r][:n]
        return tiles_to_film(out, W, H)
    dist.gather(send, gather_list=None, dst=0, group=group)
    return None


def reduce_film(film, rank, world, group=None):
    """Wide reconstruction filter: each rank passes its full-frame film of partial sums {X, Y, Z, weightSum};
    rank 0 receives their sum (ImageFilm::AddSample is a sum, film/image.cpp:96-136), other ranks get None.
    One collective."""
    if world == 1:
        return film
    buf = film.contiguous().clone()
    dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, group=group)
    return buf if rank == 0 else None


def exchange_film(film, rank, world, wide_filter=False, group=None):
    """The end-of-frame film exchange: gather of owned tiles (default box filter) or sum-reduce (wide filter)."""
    return reduce_film(film, rank, world, group) if wide_filter else gather_film(film, rank, world, group)
