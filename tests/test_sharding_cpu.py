"""Multi-rank path on CPU: world_size-2 gloo processes shard frames as
dataset/sk_dataloader.py:196-198 does and exchange per-frame arrays with ONE all-gather; the
gathered bank must equal the single-process one."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _frames(n_frames):
    rng = np.random.default_rng(0)
    return [torch.from_numpy(rng.random((50 + 7 * f, 19)).astype(np.float32)) for f in range(n_frames)]


def _worker(rank, world, port, n_frames, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lidal_amd.score import frame_range, gather_frames
    frames = _frames(n_frames)
    mine = {f: frames[f] for f in frame_range(n_frames, world, rank)}
    got = gather_frames(mine, n_frames)
    ok = len(got) == n_frames and all(torch.equal(a, b) for a, b in zip(got, frames))
    worlds = gather_frames({f: frames[f][:, :3].double() for f in mine}, n_frames)
    ok = ok and all(torch.equal(a, b[:, :3].double()) for a, b in zip(worlds, frames))
    torch.save(ok, os.path.join(out_dir, 'ok_%d.pt' % rank))
    dist.destroy_process_group()


def test_frame_range_is_the_reference_contiguous_split():
    from lidal_amd.score import frame_range
    assert [list(frame_range(10, 4, r)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert [len(frame_range(256, 8, r)) for r in range(8)] == [32] * 8
    assert list(frame_range(3, 8, 5)) == []
    cover = sorted(f for r in range(3) for f in frame_range(17, 3, r))
    assert cover == list(range(17))


def test_gather_frames_world_size_2_gloo(tmp_path):
    for n_frames in (7, 2):
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
        assert all(torch.load(os.path.join(str(tmp_path), 'ok_%d.pt' % r)) for r in range(2))


def test_gather_frames_single_process_is_identity():
    from lidal_amd.score import gather_frames
    frames = _frames(4)
    got = gather_frames({f: frames[f] for f in range(4)}, 4)
    assert all(a is b for a, b in zip(got, frames))
