"""N>1 path on CPU: world_size-2 gloo run of the sample split and the one-reduce read-out."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_2_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "combined ok" in p.stdout


def test_split_covers_every_sample_once():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            off, stride = distributed.rank_sample_split(r, world)
            seen += [off + k * stride for k in range(5)]
        assert sorted(seen) == list(range(5 * world))
    with pytest.raises(ValueError):
        distributed.rank_sample_split(2, 2)


def test_combine_single_process_is_weighted_mean():
    import torch
    mean = torch.rand(10, 3)
    cnt = torch.arange(1, 11, dtype=torch.int32)
    out, total = distributed.combine_running_means(mean, cnt)
    assert np.allclose(out.numpy(), mean.numpy(), atol=1e-6) and (total.numpy() == cnt.numpy()).all()
    assert distributed.resolve_rgb8(np.array([[0.999, 0.5, 1.0]], dtype=np.float32)).tolist() == [[254, 127, 255]]


def test_tile_split_partitions_the_image():
    w, h = 70, 50                                    # not a multiple of the tile size
    for world in (1, 2, 3, 8):
        owners = np.zeros(w * h, dtype=np.int32)
        for r in range(world):
            ts, off, stride = distributed.rank_tile_split(r, world, 16)
            owners += distributed.tile_owner_mask(w, h, ts, off, stride)
        assert (owners == 1).all()
    m = distributed.tile_owner_mask(w, h, 16, 1, 2).reshape(h, w)
    assert not m[0, 0] and m[0, 16] and m[0, 31] and not m[0, 32]      # tiles 0,1,2 of the first row
    assert m[16, 0] == (((1 * 5 + 0) % 2) == 1)                         # 5 tiles per row
