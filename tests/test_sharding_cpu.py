"""Multi-rank path on CPU: world_size-2 gloo processes shard frames as
dataset/sk_dataloader.py:196-198 does and exchange per-frame arrays with ONE all-gather; the
gathered bank must equal the single-process one.  The return leg -- per-supervoxel results to
rank 0, scatter into the global arrays (LiDAL.py:208-218), selection (:225-330) -- is checked
against the flags the reference's own __main__ wrote (tests/golden/selection_small.npz)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _frames(n_frames):
    rng = np.random.default_rng(0)
    return [torch.from_numpy(rng.random((50 + 7 * f, 19)).astype(np.float32)) for f in range(n_frames)]


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker(rank, world, port, n_frames, out_dir):
    _init(rank, world, port)
    from lidal_amd.score import frame_range, gather_frames
    frames = _frames(n_frames)
    mine = {f: frames[f] for f in frame_range(n_frames, world, rank)}
    got = gather_frames(mine, n_frames, (19,), torch.float32)
    ok = len(got) == n_frames and all(torch.equal(a, b) for a, b in zip(got, frames))
    worlds = gather_frames({f: frames[f][:, :3].double() for f in mine}, n_frames, (3,), torch.float64)
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
    # 7 and 2 frames: uneven / even blocks; 1 frame: rank 1 owns NOTHING and still has to join the
    # collectives with tensors of the agreed shape, dtype and device (9 frames on 8 GPUs in config 4)
    for n_frames in (7, 2, 1):
        port = _free_port()
        mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
        assert all(torch.load(os.path.join(str(tmp_path), 'ok_%d.pt' % r)) for r in range(2))


def test_gather_frames_single_process_is_identity():
    from lidal_amd.score import gather_frames
    frames = _frames(4)
    got = gather_frames({f: frames[f] for f in range(4)}, 4, (19,), torch.float32)
    assert all(a is b for a, b in zip(got, frames))


def _select_worker(rank, world, port, out_dir):
    """Each rank holds the per-frame results of ITS frame block of every sequence (as
    score_sequence returns them); rank 0 must end up with the reference's flags."""
    _init(rank, world, port)
    from lidal_amd.score import ScoreBoard, collect_sequence, frame_range
    g = np.load(os.path.join(GOLDEN, 'selection_small.npz'))
    n_frames, n_sv = int(g['n_frames']), int(g['n_sv'])
    n_seq = g['flags_in'].size // (n_frames * n_sv)
    board = ScoreBoard(g['flags_in'].size) if rank == 0 else None
    for s_i in range(n_seq):
        mine = list(frame_range(n_frames, world, rank))
        scores, ids, ptrs = [], [], []
        for f in mine:
            lo = (s_i * n_frames + f) * n_sv
            sl = slice(lo, lo + n_sv)
            scores.append((torch.from_numpy(g['sv_interds'][sl]), torch.from_numpy(g['sv_interes'][sl]),
                           torch.from_numpy(g['sv_centers_local'][sl])))
            ids.append(np.arange(lo, lo + n_sv, dtype=np.int64))
            ptrs.append(torch.from_numpy(np.concatenate([[0], np.cumsum(g['sv_pnums'][sl])])))
        frames = collect_sequence(scores, ids, ptrs, mine[0] if mine else 0, n_frames)
        assert (frames is None) == (rank != 0)
        if rank == 0:
            board.add_sequence(s_i, frames)
    ok = True
    if rank == 0:
        ok = (np.array_equal(board.sv_interds, g['sv_interds'])
              and np.array_equal(board.sv_interes, g['sv_interes'])
              and np.array_equal(board.sv_pnums, g['sv_pnums'])
              and np.array_equal(board.sv_centers, g['sv_centers']))      # incl. the +1000*seq offset
        flags = board.select(g['flags_in'], int(g['train_point_num']))
        ok = ok and np.array_equal(flags, g['flags_out'])
    torch.save(bool(ok), os.path.join(out_dir, 'sel_%d.pt' % rank))
    dist.destroy_process_group()


def test_rank0_collection_and_selection_match_reference_flags_gloo(tmp_path):
    """sv results of all ranks -> rank 0 -> global arrays -> select() equals the
    sv_flag files /root/reference/score/sv_level/LiDAL.py's __main__ wrote (10 sequences x 25
    frames x 20 supervoxels), with the frames of every sequence split over 2 ranks."""
    port = _free_port()
    mp.spawn(_select_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), 'sel_%d.pt' % r)) for r in range(2))


def test_collection_single_process_matches_reference_flags():
    from lidal_amd.score import ScoreBoard, collect_sequence
    g = np.load(os.path.join(GOLDEN, 'selection_small.npz'))
    n_frames, n_sv = int(g['n_frames']), int(g['n_sv'])
    board = ScoreBoard(g['flags_in'].size)
    for s_i in range(g['flags_in'].size // (n_frames * n_sv)):
        scores, ids, ptrs = [], [], []
        for f in range(n_frames):
            lo = (s_i * n_frames + f) * n_sv
            sl = slice(lo, lo + n_sv)
            scores.append((torch.from_numpy(g['sv_interds'][sl]), torch.from_numpy(g['sv_interes'][sl]),
                           torch.from_numpy(g['sv_centers_local'][sl])))
            ids.append(np.arange(lo, lo + n_sv, dtype=np.int64))
            ptrs.append(torch.from_numpy(np.concatenate([[0], np.cumsum(g['sv_pnums'][sl])])))
        board.add_sequence(s_i, collect_sequence(scores, ids, ptrs, 0, n_frames))
    assert np.array_equal(board.sv_centers, g['sv_centers'])
    assert np.array_equal(board.select(g['flags_in'], int(g['train_point_num'])), g['flags_out'])


def test_needed_frames_is_the_union_of_the_neighbour_windows():
    from lidal_amd.score import frame_range, needed_frames, neighbour_ids
    for n, world, nei in ((13, 2, 10), (40, 3, 24), (256, 8, 10), (4, 3, 2)):
        seen = set()
        for r in range(world):
            need = needed_frames(n, world, r, nei)
            want = set()
            for i in frame_range(n, world, r):
                want |= {i} | set(neighbour_ids(i, n, nei))
            assert need == sorted(want)
            seen |= set(frame_range(n, world, r))
            # bounded: a block plus its halo plus the wrap-rule frames -- never the whole long sequence
            assert len(need) <= len(frame_range(n, world, r)) + 2 * nei
        assert seen == set(range(n))
    assert len(needed_frames(4541, 8, 3, 24)) == 568 + 24          # SemanticKITTI seq 00 on 8 GPUs


def _halo_worker(rank, world, port, n_frames, nei, out_dir):
    _init(rank, world, port)
    from lidal_amd.score import HaloExchange, frame_range, needed_frames
    frames = _frames(n_frames)
    mine = list(frame_range(n_frames, world, rank))
    probs = {f: frames[f] for f in mine}
    worlds = {f: frames[f][:, :3].double() for f in mine}
    hx = HaloExchange(n_frames, nei, {f: frames[f].shape[0] for f in mine})
    hx.exchange('world', (3,), torch.float64, worlds)
    hx.exchange('prob', (19,), torch.float32, {f: probs[f] for f in hx.exports})
    have = hx.finish({'world': worlds, 'prob': probs})
    need = needed_frames(n_frames, world, rank, nei)
    ok = sorted(have['prob']) == need and sorted(have['world']) == need
    ok = ok and all(torch.equal(have['prob'][f], frames[f]) for f in need)
    ok = ok and all(torch.equal(have['world'][f], frames[f][:, :3].double()) for f in need)
    torch.save(bool(ok), os.path.join(out_dir, 'halo_%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_delivers_exactly_the_needed_frames_gloo(tmp_path):
    """The bounded-memory exchange (score/sharding.py HaloExchange): every rank ends up with its block,
    its halo and the wrap-rule frames -- bit-equal to the originals, nothing else -- for 2 and 3 ranks,
    windows 10 and 24, and a rank that owns no frame at all."""
    for world, n_frames, nei in ((2, 13, 10), (2, 27, 24), (3, 40, 10), (3, 4, 2)):
        port = _free_port()
        mp.spawn(_halo_worker, args=(world, port, n_frames, nei, str(tmp_path)), nprocs=world, join=True)
        assert all(torch.load(os.path.join(str(tmp_path), 'halo_%d.pt' % r)) for r in range(world)), (world, n_frames)


# ---- lidal_amd.data_parallel.DataParallel (train.py:49-53's DistributedDataParallel for the planned step) ----------
class _FlatGrads(torch.autograd.Function):
    """Stands in for the planned step (network/plan.py): one node whose parameter gradients are views of ONE buffer."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return x @ w + b

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        flat = torch.empty(w.numel() + b_numel(w), dtype=torch.float32)
        gw, gb = flat[:w.numel()].view_as(w), flat[w.numel():]
        gw.copy_(x.t() @ g)
        gb.copy_(g.sum(0))
        return g @ w.t(), gw, gb


def b_numel(w):
    return w.shape[1]


class _Net(torch.nn.Module):
    def __init__(self, flat):
        super().__init__()
        g = torch.Generator().manual_seed(3)
        self.w = torch.nn.Parameter(torch.randn(8, 4, generator=g))
        self.b = torch.nn.Parameter(torch.randn(4, generator=g))
        self.register_buffer('seen', torch.zeros(1, dtype=torch.int64))
        self.flat = flat

    def forward(self, x):
        return _FlatGrads.apply(x, self.w, self.b) if self.flat else x @ self.w + self.b


def _dp_worker(rank, world, port, out_dir):
    _init(rank, world, port)
    from lidal_amd.data_parallel import DataParallel
    ok = True
    for flat in (True, False):
        net = _Net(flat)
        if rank == 1:                           # rank 0's state must win
            with torch.no_grad():
                net.w.add_(1.0)
                net.seen.add_(5)
        dp = DataParallel(net)
        ref = _Net(flat)
        ok = ok and torch.equal(net.w, ref.w) and int(net.seen) == 0
        grads = []
        for r in range(world):                  # every rank's single-process gradient, computed locally
            m = _Net(False)
            x = torch.randn(16, 8, generator=torch.Generator().manual_seed(10 + r))
            m(x).square().sum().backward()
            grads.append((m.w.grad.clone(), m.b.grad.clone()))
        x = torch.randn(16, 8, generator=torch.Generator().manual_seed(10 + rank))
        for step in range(2):                   # (the hook re-arms itself)
            dp.zero_grad()
            dp(x).square().sum().backward()
            ok = ok and torch.allclose(net.w.grad, sum(g[0] for g in grads) / world, rtol=1e-6, atol=1e-6)
            ok = ok and torch.allclose(net.b.grad, sum(g[1] for g in grads) / world, rtol=1e-6, atol=1e-6)
        ok = ok and dp.reductions == 2 and dp.flat_reductions == (2 if flat else 0)
        before = net.w.grad.clone()             # autograd.grad through the outputs accumulates nothing: nothing to reduce
        torch.autograd.grad(dp(x).square().sum(), [net.w])
        ok = ok and dp.reductions == 2 and torch.equal(net.w.grad, before)
        ok = ok and list(dp.state_dict()) == ['module.w', 'module.b', 'module.seen']
    torch.save(ok, os.path.join(out_dir, 'dp_ok_%d.pt' % rank))
    dist.destroy_process_group()


def test_data_parallel_wrapper_world_size_2_gloo(tmp_path):
    """Gradients = mean over the ranks (one in-place collective when they lie back to back in one buffer, a flattened
    copy otherwise), rank 0's parameters and buffers broadcast at construction, DDP's 'module.' state_dict prefix."""
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), 'dp_ok_%d.pt' % r)) for r in range(2))
