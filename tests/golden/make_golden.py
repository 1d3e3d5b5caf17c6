"""Generate the golden fixtures under tests/golden/.  Runs ONLY in the build container: it
imports the reference's own Python files from /root/reference (which do not exist on the GPU
box) on top of the CPU oracle, and commits inputs + expected outputs as small .npz/.json data.

  python tests/golden/make_golden.py            # all fixtures
  python tests/golden/make_golden.py model      # just one group: model | scoring | selection

What each fixture pins
  model_small.npz     reference network/spvcnn.py + network/minkunet.py (unchanged files) run on
                      oracle.tsref: logits/feat for a seeded ~3 k-point scan (eval), one train
                      step (loss + gradient norms), every kernel map the forward builds.
  model_10k.npz       the same SPVCNN file on a seeded 10 k-point scan (BASELINE.json configs[0]).
  state_dict_*.json   the checkpoint compatibility surface (keys, shapes, dtypes).
  scoring_small.npz   reference score/sv_level/LiDAL.py::worker_func run unchanged on synthetic
                      frames; also asserts oracle.scoring_ref reproduces it (the oracle PIN).
  selection_small.npz reference score/sv_level/LiDAL.py __main__ run unchanged in a temp
                      Processing_files tree (10 sequences); asserts oracle.scoring_ref.select
                      reproduces the written sv_flag files.
  io_small.npz        raw bytes of files of that tree (written by the reference's __main__, or
                      consumed by it) + the arrays they decode to: fixture of lidal_amd.io.
"""
import json
import os
import pickle
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from weights import fill_state_dict, state_dict_signature  # noqa: E402


def _import_reference_models():
    from oracle.install import install_as_torchsparse
    ts = install_as_torchsparse()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from network.spvcnn import SPVCNN
    from network.minkunet import MinkUNet
    return ts, SPVCNN, MinkUNet


def make_model():
    ts, SPVCNN, MinkUNet = _import_reference_models()
    from lidal_amd import synth
    torch.set_num_threads(8)
    batch = synth.make_train_batch(n_frames=2, n_points=1600, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b'])
    feats = torch.from_numpy(batch['feats_v_b'])
    labels = torch.from_numpy(batch['labels_v_b'])
    out = {'coords': batch['coords_v_b'], 'feats': batch['feats_v_b'],
           'labels': batch['labels_v_b']}
    for name, cls in (('spvcnn', SPVCNN), ('minkunet', MinkUNet)):
        model = fill_state_dict(cls(19))
        with open(os.path.join(HERE, 'state_dict_%s.json' % name), 'w') as f:
            json.dump(state_dict_signature(model), f)
        model.eval()
        x = ts.SparseTensor(feats.clone(), coords.clone())
        with torch.no_grad():
            logits, feat = model(x)
        out[name + '_logits'] = logits.numpy()
        out[name + '_feat'] = feat.numpy()
        # PIN of oracle/models_ref.py (the architecture restatement that travels to the GPU box)
        from oracle import models_ref
        ref_cls = {'spvcnn': models_ref.SPVCNNRef, 'minkunet': models_ref.MinkUNetRef}[name]
        port = fill_state_dict(ref_cls(19)).eval()
        with torch.no_grad():
            l2, f2 = port(ts.SparseTensor(feats.clone(), coords.clone()))
        assert torch.equal(l2, logits) and torch.equal(f2, feat), 'models_ref != reference files'
        if name == 'minkunet':                      # every kernel map of the U-Net
            for key, km in x.kmaps.items():
                tag = 's%d_k%d_c%d' % (key[0][0], key[1][0], key[2][0])
                out['kmap_%s_nbmaps' % tag] = km[0].numpy().astype(np.int32)
                out['kmap_%s_nbsizes' % tag] = km[1].numpy().astype(np.int32)
            for s, c in x.cmaps.items():
                out['cmap_s%d' % s[0]] = c.numpy().astype(np.int32)
        # one training step (train.py:127-140); dropout off so CPU and GPU RNG cannot differ.
        # Run in float64: through 49 conv + train-mode BN layers the f32 CPU oracle's own gradients
        # deviate from its f64 run by up to 2e-3 (stem), i.e. more than the 1e-4 bar being checked,
        # so the f64 run is the golden (the f32 loss is kept to show the two agree).
        gkeys = ['stem.0.kernel', 'stage2.1.net.0.kernel', 'stage4.2.net.3.kernel',
                 'up1.0.net.0.kernel', 'up4.1.1.net.3.kernel', 'classifier.0.weight',
                 'stage1.0.net.1.weight']
        out[name + '_grad_keys'] = np.array(gkeys)
        for dt, tag in ((torch.float32, '_f32'), (torch.float64, '')):
            model = fill_state_dict(cls(19)).to(dt)
            model.train()
            if hasattr(model, 'dropout'):
                model.dropout.p = 0.0
            opt = torch.optim.Adam(model.parameters())
            opt.zero_grad()
            logits, _ = model(ts.SparseTensor(feats.clone().to(dt), coords.clone()))
            loss = torch.nn.functional.cross_entropy(logits, labels, ignore_index=255,
                                                     reduction='mean')
            loss.backward()
            named = dict(model.named_parameters())
            # '' = float64 golden; '_f32' = the f32 CPU oracle, kept to calibrate the test's bars
            out[name + '_train_loss' + tag] = np.float64(loss.item())
            out[name + '_grad_norms' + tag] = np.array(
                [named[k].grad.norm().item() for k in gkeys], dtype=np.float64)
            out[name + '_grad_stem' + tag] = named['stem.0.kernel'].grad.numpy().astype(np.float32)
            if tag == '':
                # whole gradient tensors of the sampled parameters (kernels wider than 64 channels
                # cut to their leading 32 x 32 block): the bf16 test needs directions, not norms
                for i, k in enumerate(gkeys + (['point_transforms.1.0.weight'] if name == 'spvcnn' else [])):
                    gr = named[k].grad.numpy()
                    if gr.ndim == 3 and max(gr.shape[1:]) > 64:
                        gr = gr[:, :32, :32]
                    out['%s_gradfull_%d' % (name, i)] = gr.astype(np.float32).copy()
                    out['%s_gradfull_key_%d' % (name, i)] = np.array(k)
            out[name + '_grad_up1dc' + tag] = (named['up1.0.net.0.kernel'].grad.numpy()[:, :8, :8]
                                               .astype(np.float32).copy())
        out[name + '_train_logits'] = logits.detach().numpy().astype(np.float32)
        print(name, 'loss', loss.item(), 'logits', tuple(logits.shape))
    np.savez_compressed(os.path.join(HERE, 'model_small.npz'), **out)


def make_model_10k():
    """BASELINE.json configs[0] as written: a synthetic 10 k-point scan, 0.05 m voxels, SPVCNN
    forward on the CPU path -- the reference's network/spvcnn.py (unchanged) on the oracle."""
    ts, SPVCNN, _ = _import_reference_models()
    from lidal_amd import synth
    torch.set_num_threads(8)
    batch = synth.make_train_batch(n_frames=1, n_points=10000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b'])
    feats = torch.from_numpy(batch['feats_v_b'])
    model = fill_state_dict(SPVCNN(19)).eval()
    with torch.no_grad():
        logits, feat = model(ts.SparseTensor(feats.clone(), coords.clone()))
    np.savez_compressed(os.path.join(HERE, 'model_10k.npz'), coords=batch['coords_v_b'],
                        feats=batch['feats_v_b'], spvcnn_logits=logits.numpy(),
                        spvcnn_feat_sample=feat.numpy()[::16].copy())
    print('model_10k:', tuple(coords.shape), 'voxels; logits', tuple(logits.shape))


def _stub_nuscenes(tmp):
    d = os.path.join(tmp, 'stubs', 'nuscenes', 'utils')
    os.makedirs(d)
    open(os.path.join(tmp, 'stubs', 'nuscenes', '__init__.py'), 'w').close()
    open(os.path.join(d, '__init__.py'), 'w').close()
    with open(os.path.join(d, 'splits.py'), 'w') as f:
        f.write('def create_splits_scenes():\n    return {"train": []}\n')
    return os.path.join(tmp, 'stubs')


def _synthetic_probs(rng, world, n_classes=19):
    """Spatially coherent class probabilities: a smooth function of world position + noise,
    so neighbouring frames mostly agree (realistic KL / entropy ranges)."""
    w = rng.standard_normal((3, n_classes)) * 0.35
    logit = world @ w + rng.standard_normal((world.shape[0], n_classes)) * 0.7
    logit = logit - logit.max(1, keepdims=True)
    p = np.exp(logit)
    return (p / p.sum(1, keepdims=True)).astype(np.float32)


def make_scoring():
    from lidal_amd import synth
    from oracle import scoring_ref
    from sklearn.neighbors import KDTree
    n_frames, P, nei_num, dis = 27, 400, 24, 0.1
    frames = synth.make_sequence(n_frames, n_points=P, seed=11, step=0.02, n_beams=16, n_az=128)
    rng = np.random.default_rng(5)
    # thin scans would never match within 0.1 m, so half of every frame re-observes a shared
    # static point set with small jitter (matches + near-ties around the threshold)
    shared = frames[0]['world'][: P // 2].copy()
    probs, worlds = [], []
    for f in frames:
        wc = f['world'].copy()
        wc[: P // 2] = shared + rng.normal(0, 0.04, size=shared.shape)
        worlds.append(wc)
        probs.append(_synthetic_probs(np.random.default_rng(99), wc))
    tmp = tempfile.mkdtemp()
    stubs = _stub_nuscenes(tmp)
    sys.path.insert(0, stubs)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import score.sv_level.LiDAL as L          # the reference file, unchanged
    prob_files, kd_files, sv_files = [], [], []
    for i, f in enumerate(frames):
        pf = os.path.join(tmp, 'prob_%06d.npy' % i)
        np.save(pf, probs[i])
        kf = os.path.join(tmp, 'kd_%06d.pickle' % i)
        with open(kf, 'wb') as fh:
            pickle.dump(KDTree(worlds[i]), fh)
        sf = os.path.join(tmp, 'sv_%06d.pickle' % i)
        with open(sf, 'wb') as fh:
            pickle.dump((f['sv_id'], f['sv2point']), fh)
        prob_files.append(pf), kd_files.append(kf), sv_files.append(sf)
    L.init_worker(False, nei_num, dis, '00', prob_files, kd_files, sv_files)
    out = {'nei_num': nei_num, 'dis_thresh': dis,
           'probs': np.stack(probs), 'worlds': np.stack(worlds),
           'sv2point': np.stack([np.stack(f['sv2point']) for f in frames])}
    ref_d, ref_e, ref_n, ref_c = [], [], [], []
    for i in range(n_frames):
        sv_id, d, e, n, c = L.worker_func(i)
        od, oe, on, oc, pd, pe = scoring_ref.score_frame(i, probs, worlds, frames[i]['sv2point'],
                                                         nei_num, dis, return_points=True)
        assert np.array_equal(d, od) and np.array_equal(e, oe), 'oracle != reference worker_func'
        assert np.array_equal(n, on) and np.array_equal(c, oc)
        ref_d.append(d), ref_e.append(e), ref_n.append(n), ref_c.append(c)
        if i == 0:
            out['interd_points_f0'] = pd
            out['intere_points_f0'] = pe
    out.update(sv_interds=np.stack(ref_d), sv_interes=np.stack(ref_e),
               sv_pnums=np.stack(ref_n), sv_centers=np.stack(ref_c))
    np.savez_compressed(os.path.join(HERE, 'scoring_small.npz'), **out)
    print('scoring: oracle == reference worker_func on', n_frames, 'frames;',
          'mean interd', float(np.mean(ref_d)), 'mean intere', float(np.mean(ref_e)))


def make_selection():
    from lidal_amd import synth
    from oracle import scoring_ref
    from sklearn.neighbors import KDTree
    seqs = ['00', '01', '02', '03', '04', '05', '06', '07', '09', '10']   # LiDAL.py:125
    n_frames, P, n_sv = 25, 160, 20
    tmp = tempfile.mkdtemp()
    stubs = _stub_nuscenes(tmp)
    base = os.path.join(tmp, 'Processing_files', 'SK')
    rng = np.random.default_rng(3)
    all_flags, all_probs, all_worlds, all_sv = [], [], [], []
    gid = 0
    for s_i, seq in enumerate(seqs):
        frames = synth.make_sequence(n_frames, n_points=P, seed=100 + s_i, step=0.6,
                                     n_beams=8, n_az=64, n_sv=n_sv)
        shared = frames[0]['world'][: P // 2].copy()
        for d in ('prob_map/SPVCNN/fr/0r', 'kdtree', 'super_voxel/KMeans', 'sv_flag/KMeans/0r'):
            os.makedirs(os.path.join(base, d, seq))
        for i, f in enumerate(frames):
            wc = f['world'].copy()
            wc[: P // 2] = shared + rng.normal(0, 0.04, size=shared.shape) + [0.6 * i, 0, 0]
            prob = _synthetic_probs(np.random.default_rng(7 + s_i), wc)
            flags = (rng.random(n_sv) < 0.1).astype(np.int64)
            flags[rng.random(n_sv) < 0.1] = 2
            sv_id = np.arange(gid, gid + n_sv, dtype=np.int64)
            gid += n_sv
            name = '%06d' % i
            np.save(os.path.join(base, 'prob_map/SPVCNN/fr/0r', seq, name + '.npy'), prob)
            with open(os.path.join(base, 'kdtree', seq, name + '.pickle'), 'wb') as fh:
                pickle.dump(KDTree(wc), fh)
            with open(os.path.join(base, 'super_voxel/KMeans', seq, name + '.pickle'), 'wb') as fh:
                pickle.dump((sv_id, f['sv2point']), fh)
            np.save(os.path.join(base, 'sv_flag/KMeans/0r', seq, name + '.npy'), flags)
            all_flags.append(flags), all_probs.append(prob), all_worlds.append(wc)
            all_sv.append(f['sv2point'])
    env = dict(os.environ, PYTHONPATH=stubs + ':' + REF)
    r = subprocess.run([sys.executable, '-m', 'score.sv_level.LiDAL', '--dataset_name', 'SK',
                        '--model_name', 'SPVCNN', '--r_id', '1'], cwd=tmp, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    new_flags = []
    for seq in seqs:
        for i in range(n_frames):
            new_flags.append(np.load(os.path.join(base, 'sv_flag/KMeans/SPVCNN/LiDAL/1r', seq,
                                                  '%06d.npy' % i)))
    sv_pnums = np.load(os.path.join(base, 'super_voxel/KMeans/sv_pnums.npy'))
    sv_centers = np.load(os.path.join(base, 'super_voxel/KMeans/sv_centers.npy'))
    # oracle: score every frame, then select
    d_all, e_all, c_all = [], [], []
    for s_i in range(len(seqs)):
        lo = s_i * n_frames
        pr, wo = all_probs[lo:lo + n_frames], all_worlds[lo:lo + n_frames]
        for i in range(n_frames):
            d, e, n, c = scoring_ref.score_frame(i, pr, wo, all_sv[lo + i], 24, 0.1)
            d_all.append(d), e_all.append(e), c_all.append(c)
            assert np.array_equal(n, sv_pnums[(lo + i) * n_sv:(lo + i + 1) * n_sv])
            assert np.array_equal(c + np.float32(s_i * 1000.0),
                                  sv_centers[(lo + i) * n_sv:(lo + i + 1) * n_sv])
    flags_in = np.concatenate(all_flags)
    sel = scoring_ref.select(flags_in, np.concatenate(d_all), np.concatenate(e_all), sv_pnums,
                             sv_centers, 2349559532)
    assert np.array_equal(sel, np.concatenate(new_flags)), 'oracle select != reference __main__'
    np.savez_compressed(os.path.join(HERE, 'selection_small.npz'), flags_in=flags_in,
                        sv_interds=np.concatenate(d_all), sv_interes=np.concatenate(e_all),
                        sv_pnums=sv_pnums, sv_centers=sv_centers,
                        sv_centers_local=np.concatenate(c_all),      # per frame, before the +1000*seq offset
                        n_frames=n_frames, n_sv=n_sv,
                        flags_out=np.concatenate(new_flags), train_point_num=2349559532)
    # io_small.npz: the BYTES of files of that tree -- the ones the reference's __main__ wrote
    # (sv_flag of round 1, sv_pnums.npy, sv_centers.npy: LiDAL.py:220-222,328-330) and the inputs it
    # consumed in the formats of Appendix B (prob .npy, supervoxel pickle, round-0 sv_flag) -- with
    # the arrays they must decode to.  Data, not source.
    def raw(*parts):
        return np.frombuffer(open(os.path.join(base, *parts), 'rb').read(), dtype=np.uint8)
    np.savez_compressed(
        os.path.join(HERE, 'io_small.npz'),
        file_sv_flag_1r=raw('sv_flag/KMeans/SPVCNN/LiDAL/1r', seqs[2], '000004.npy'),
        sv_flag_1r=new_flags[2 * n_frames + 4],
        file_sv_flag_0r=raw('sv_flag/KMeans/0r', seqs[2], '000004.npy'),
        sv_flag_0r=all_flags[2 * n_frames + 4],
        file_sv_pnums=raw('super_voxel/KMeans/sv_pnums.npy'), sv_pnums=sv_pnums,
        file_sv_centers=raw('super_voxel/KMeans/sv_centers.npy'), sv_centers=sv_centers,
        file_prob=raw('prob_map/SPVCNN/fr/0r', seqs[2], '000004.npy'),
        prob=all_probs[2 * n_frames + 4],
        file_supervoxel=raw('super_voxel/KMeans', seqs[2], '000004.pickle'),
        sv_id=np.arange((2 * n_frames + 4) * n_sv, (2 * n_frames + 5) * n_sv, dtype=np.int64),
        sv2point=np.stack(all_sv[2 * n_frames + 4]))
    print('selection: oracle == reference __main__;', int((sel == 1).sum()), 'labelled,',
          int((sel == 2).sum()), 'pseudo-labelled of', sel.size)


def make_voxelize():
    """Reference dataset/sk_dataset.py (SK_Dataset.__getitem__ + collate_fn, unchanged) on synthetic
    velodyne .bin files; pins oracle/voxelize_ref.py and gives the GPU voxeliser its fixture."""
    from lidal_amd import data as ldata
    from lidal_amd import synth
    from oracle import voxelize_ref
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.makedirs(os.path.join(tmp, 'Processing_files', 'SK'))
    os.makedirs(os.path.join(tmp, 'seq', '00', 'velodyne'))
    rng = np.random.default_rng(4)
    world = synth.make_world(9)
    files, scans = [], []
    for i in range(2):
        pts, inten = synth.raycast_scan(world, (20.0 + 3 * i, 0.0), rng, n_beams=32, n_az=256)
        f = os.path.join(tmp, 'seq', '00', 'velodyne', '%06d.bin' % i)
        np.concatenate([pts, inten[:, None]], 1).astype(np.float32).tofile(f)
        files.append(f), scans.append((pts, inten))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(tmp)
    try:
        from dataset.sk_dataset import SK_Dataset       # the reference file, unchanged
        ds = SK_Dataset(mode='score', lidar_files=files)
        out = {}
        samples = []
        for i, (pts, inten) in enumerate(scans):
            np.random.seed(100 + i)
            ref = ds[i]
            np.random.seed(100 + i)
            trans_m, rnd = ldata.draw_augmentation(np.random)
            cv, fv, ui, inv = voxelize_ref.voxelize_scan(pts, inten, trans_m, rnd)
            assert np.array_equal(cv, ref['coords_v']) and np.array_equal(inv, ref['inverse_idxs'])
            assert np.array_equal(fv, ref['feats_v']), 'oracle voxelize != reference __getitem__'
            out.update({'points%d' % i: pts, 'intensity%d' % i: inten, 'trans_m%d' % i: trans_m,
                        'rnd%d' % i: rnd, 'coords_v%d' % i: ref['coords_v'],
                        'feats_v%d' % i: ref['feats_v'], 'inverse%d' % i: ref['inverse_idxs']})
            samples.append(ref)
        col = ds.collate_fn(samples)
        out.update(coords_v_b=col['coords_v_b'].numpy(), feats_v_b=col['feats_v_b'].numpy(),
                   inverse_indices_b=col['inverse_indices_b'].numpy())
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'voxelize_small.npz'), **out)
    print('voxelize: oracle == reference SK_Dataset on 2 scans;', out['coords_v_b'].shape)


def make_register():
    """Reference dataset/prepare_kdtree_sk.py (parse_calibration, parse_poses, process_frame,
    unchanged) on a synthetic calib/poses/velodyne triple: world coordinates held by the pickled
    KDTree are the fixture for lidal_amd.data.register_scan."""
    from lidal_amd import synth
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.makedirs(os.path.join(tmp, 'Processing_files', 'SK', 'kdtree', '00'))
    os.makedirs(os.path.join(tmp, 'sequences', '00', 'velodyne'))
    rng = np.random.default_rng(8)
    pts, inten = synth.raycast_scan(synth.make_world(2), (15.0, 0.0), rng, n_beams=16, n_az=128)
    fbin = os.path.join(tmp, 'sequences', '00', 'velodyne', '000003.bin')
    np.concatenate([pts, inten[:, None]], 1).astype(np.float32).tofile(fbin)
    calib_txt = 'P0: ' + ' '.join(['1'] * 12) + '\nTr: 4.27e-04 -9.99e-01 -8.08e-03 -1.19e-02 -7.21e-03 8.08e-03 -9.99e-01 -5.40e-02 9.99e-01 4.85e-04 -7.20e-03 -2.92e-01\n'
    open(os.path.join(tmp, 'calib.txt'), 'w').write(calib_txt)
    th = 0.31
    pose_vals = [np.cos(th), 0.02, np.sin(th), 12.5, -0.01, 1.0, 0.03, -0.7, -np.sin(th), 0.01, np.cos(th), 101.25]
    open(os.path.join(tmp, 'poses.txt'), 'w').write(' '.join('%.9e' % v for v in pose_vals) + '\n')
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(tmp)
    try:
        import dataset.prepare_kdtree_sk as K            # the reference file, unchanged
        calib = K.parse_calibration(os.path.join(tmp, 'calib.txt'))
        poses = K.parse_poses(os.path.join(tmp, 'poses.txt'), calib)
        K.process_frame(0, [fbin], poses)
        with open(os.path.join(tmp, 'Processing_files/SK/kdtree/00/000003.pickle'), 'rb') as fh:
            tree = pickle.load(fh)
    finally:
        os.chdir(cwd)
    from lidal_amd import data as ldata
    c2 = ldata.parse_calibration(os.path.join(tmp, 'calib.txt'))
    p2 = ldata.parse_poses(os.path.join(tmp, 'poses.txt'), c2)
    assert np.array_equal(p2[0], poses[0]) and np.array_equal(c2['Tr'], calib['Tr'])
    np.savez_compressed(os.path.join(HERE, 'register_small.npz'), points=pts, pose=poses[0],
                        world=np.asarray(tree.data), calib_txt=calib_txt,
                        poses_txt=open(os.path.join(tmp, 'poses.txt')).read())
    print('register: fixture from reference process_frame,', np.asarray(tree.data).shape)


if __name__ == '__main__':
    assert os.path.isdir(REF), 'make_golden.py needs /root/reference (build container only)'
    which = sys.argv[1:] or ['model', 'model_10k', 'scoring', 'selection', 'voxelize', 'register']
    if 'model' in which:
        make_model()
    if 'model_10k' in which:
        make_model_10k()
    if 'scoring' in which:
        make_scoring()
    if 'selection' in which:
        make_selection()
    if 'voxelize' in which:
        make_voxelize()
    if 'register' in which:
        make_register()
