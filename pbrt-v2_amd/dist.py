"""Multi-GPU: pixel-tile sharding + the film gather.

The path shards with NO data-path collective: the scene is replicated in each GPU's HBM, rank r
renders the 32x32 super-tiles t with t % world == r (hpt_render_desc.shard_rank/shard_count —
the same round-robin the kernel's work counter walks), and the only communication is ONE gather
of film tiles (with a one-pixel apron: round 4) to rank 0 at end of frame: 16 B/pixel, 37 MB at 1080p, point-to-point over xGMI
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


TW = TILE + 2          # a tile record: the tile and a one-pixel apron (csrc/hpt_multi.hip, HPT_TILE_W)
_DIRS = [(-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]   # left, right, up, down, the diagonals: the carrier's priority order


def apron_carrier(x_count, y_count, rank, world):
    """(H, W) int32: for every film pixel q that lies in a tile NOT owned by `rank`, the index of the one tile of `rank` whose record carries q
    in its apron (-1: none) — apron_carrier_is of csrc/hpt_multi.hip: the first of q's tile's neighbours left, right, up, down, up-left,
    up-right, down-left, down-right whose apron contains q and which `rank` owns.  Why: a camera sample with an exactly integer image
    coordinate also lands in the neighbouring pixel (film/image.cpp:82-89); across a tile border between shards that contribution sits in
    the rendering shard's film outside its own tiles, and a gather of bare 32 x 32 tiles would lose it."""
    nx, ny = tile_grid(x_count, y_count)
    yy, xx = np.mgrid[0:y_count, 0:x_count]
    bx, by, u, v = xx >> 5, yy >> 5, xx & 31, yy & 31
    own = ((by * nx + bx) % world) == rank
    car = np.full((y_count, x_count), -1, dtype=np.int32)
    for dx, dy in _DIRS:
        ok = np.ones_like(own)
        if dx < 0: ok &= u == 0
        if dx > 0: ok &= u == 31
        if dy < 0: ok &= v == 0
        if dy > 0: ok &= v == 31
        tx, ty = bx + dx, by + dy
        ok &= (tx >= 0) & (ty >= 0) & (tx < nx) & (ty < ny)
        t = ty * nx + tx
        ok &= (t % world) == rank
        ok &= ~own & (car < 0)
        car[ok] = t[ok]
    return car


def film_to_records(film, rank, world):
    """film (H, W, 4) of shard `rank` -> (n_owned, TW*TW*4): its tiles with their aprons, in the order t = rank, rank + world, ..."""
    H, W, _ = film.shape
    nx, ny = tile_grid(W, H)
    f = film.detach().cpu().numpy() if isinstance(film, torch.Tensor) else np.asarray(film)
    car = apron_carrier(W, H, rank, world)
    pad = np.zeros((ny * TILE + 2, nx * TILE + 2, 4), dtype=f.dtype)
    pad[1:H + 1, 1:W + 1] = f
    cpad = np.full((ny * TILE + 2, nx * TILE + 2), -2, dtype=np.int32)
    cpad[1:H + 1, 1:W + 1] = car
    recs = []
    for t in range(rank, nx * ny, world):
        x0, y0 = (t % nx) * TILE, (t // nx) * TILE
        w = pad[y0:y0 + TW, x0:x0 + TW].copy()
        keep = cpad[y0:y0 + TW, x0:x0 + TW] == t
        keep[1:TILE + 1, 1:TILE + 1] = True
        w[~keep] = 0
        recs.append(w.reshape(-1))
    out = np.stack(recs) if recs else np.zeros((0, TW * TW * 4), dtype=f.dtype)
    return torch.from_numpy(out).to(film.device) if isinstance(film, torch.Tensor) else out


def add_records(film_np, recs, rank, world):
    """the root's half: ADD the records of shard `rank` into the frame (hpt_unpack_tiles_kernel)"""
    H, W, _ = film_np.shape
    nx, ny = tile_grid(W, H)
    pad = np.zeros((ny * TILE + 2, nx * TILE + 2, 4), dtype=film_np.dtype)
    for k, t in enumerate(range(rank, nx * ny, world)):
        x0, y0 = (t % nx) * TILE, (t // nx) * TILE
        pad[y0:y0 + TW, x0:x0 + TW] += recs[k].reshape(TW, TW, 4)
    film_np += pad[1:H + 1, 1:W + 1]


def gather_film(film, rank, world, group=None):
    """Each rank passes its full-frame film tensor (its own tiles, plus the few boundary samples that spilled into neighbouring tiles);
    rank 0 receives the assembled frame, other ranks get None.  One collective; tile records (with aprons) only."""
    if world == 1:
        return film
    H, W, _ = film.shape
    nx, ny = tile_grid(W, H)
    nt = nx * ny
    per = (nt + world - 1) // world
    mine = film_to_records(film, rank, world)
    send = torch.zeros((per, TW * TW * 4), dtype=film.dtype, device=film.device)
    send[:mine.shape[0]] = mine
    if rank == 0:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=recv, dst=0, group=group)
        out = film.detach().cpu().numpy().copy()        # the root's own tiles and its own spills are in place
        for r in range(1, world):
            add_records(out, recv[r].cpu().numpy(), r, world)
        return torch.from_numpy(out).to(film.device)
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
