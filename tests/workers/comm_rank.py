"""One rank of tests/test_gpu_multi.py::test_hpt_comm_runs_with_two_processes_over_the_host_transport.

    python tests/workers/comm_rank.py <rank> <world> <dir> <case> <wide:0|1> <frames>

Every rank is a PROCESS of its own on device 0 (HPT_COMM_TRANSPORT=host: the library's pack kernel, tile bookkeeping and unpack / sum, the
hop between the processes through POSIX shared memory instead of ncclSend / ncclRecv — RCCL refuses two ranks on one device).  Rank 0
writes the communicator id to <dir>/id and the gathered frames to <dir>/frame<k>.npy; the others read the id."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
hpt = importlib.import_module("pbrt-v2_amd.hpt")
from tests.util import hash_rd, load_case  # noqa: E402


def main():
    rank, world, d, case, wide, frames = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    import torch   # device memory only
    idf = os.path.join(d, "id")

    def bcast(uid):
        if rank == 0:
            with open(idf + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idf + ".tmp", idf)
            return uid
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no communicator id" % rank)
            time.sleep(0.01)
        return open(idf, "rb").read()

    comm = hpt.Comm(rank, world, 0, bcast)
    s = load_case(case)
    dev = hpt.DeviceScene(s, 0)
    flt = abi.make_filter("gaussian") if wide else None
    if flt is not None:
        dev.set_filter(flt)
    for k in range(frames):
        rd = hash_rd(s, seed=4 + k)
        rd.shard_rank, rd.shard_count = rank, world
        film = torch.zeros((rd.y_count, rd.x_count, 4), dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        dev.render_device(s.camera, rd, film.data_ptr(), stream)
        comm.exchange_film(rd, film.data_ptr(), stream, wide_filter=bool(wide))
        torch.cuda.synchronize()
        if rank == 0:
            np.save(os.path.join(d, "frame%d.npy" % k), film.cpu().numpy())
    comm.close()
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
