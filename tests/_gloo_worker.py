"""Worker of tests/test_multi_gpu_split.py: one rank of the sample-split render on CPU (gloo).
The per-rank render is done by the oracle here (no GPU in the CPU suite); the split and the
read-out code are the product's (pathtracer_amd/distributed.py), exactly what bench.py runs over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.binding import OracleScene                    # noqa: E402
from pathtracer_amd import capi, distributed, scenes      # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h, spp_per_rank = 32, 24, 2
    s = scenes.cornell_c1(w, h)
    orc = OracleScene(s)
    off, stride = distributed.rank_sample_split(rank, world)
    mean, cnt, _, _ = orc.render(capi.make_params(w, h, seed=11, sample_offset=off, sample_stride=stride), spp_per_rank, threads=2)
    out = distributed.combine_running_means(torch.from_numpy(mean), torch.from_numpy(cnt.astype(np.int32)), dist, 0)
    ok = True
    if rank == 0:
        comb, total = out
        full, cf, rgb, _ = orc.render(capi.make_params(w, h, seed=11), spp_per_rank * world, threads=2)
        ok = bool(np.allclose(comb.numpy(), full, atol=2e-6)) and bool((total.numpy() == cf).all())
        d = np.abs(distributed.resolve_rgb8(comb.numpy()).astype(int) - rgb.astype(int))
        ok = ok and int(d.max()) <= 1
    else:
        ok = out is None
    # image-tile partition: each rank contributes only the pixels of its tiles (count 0 elsewhere); same read-out
    ts, toff, tstride = distributed.rank_tile_split(rank, world, 8)
    own = distributed.tile_owner_mask(w, h, ts, toff, tstride)
    full, cf, rgb, _ = orc.render(capi.make_params(w, h, seed=11), spp_per_rank, threads=2)
    tmean, tcnt = full * own[:, None], (cf * own).astype(np.int32)
    out2 = distributed.combine_running_means(torch.from_numpy(tmean.astype(np.float32)), torch.from_numpy(tcnt), dist, 0)
    if rank == 0:
        comb2, total2 = out2
        ok = ok and bool(np.allclose(comb2.numpy(), full, atol=2e-6)) and bool((total2.numpy() == cf).all())
        print("combined ok" if ok else "MISMATCH", flush=True)
    else:
        ok = ok and out2 is None
    flag = torch.tensor([0 if ok else 1])
    dist.all_reduce(flag)
    dist.destroy_process_group()
    sys.exit(int(flag.item()))


if __name__ == "__main__":
    main()
