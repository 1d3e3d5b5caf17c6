"""Where a phase of conv_lean_kernel spends its time (round 5, verdict item 3a).

Needs the instrumented library:  python scripts/build_variant.py stamps conv_img.hip -DLIDAL_PHASE_STAMPS
  LIDAL_AMD_LIB=scripts/_abl/lib_stamps.so python scripts/exp/phase_stamps.py [out.json]

Three layers of the 5-scan bench batch: 96->96 at stride 1 (the roofline layer), 256->256 at stride 8 (the lean kernel,
not its deep form: forced), 32->32 at stride 2.  Per layer: launch time (events, 20 launches, stamps written too: the
product kernel's time beside it comes from scripts/exp_img.py), then from ONE instrumented launch the per-wave sums of
the five intervals of a phase (csrc/conv_img.hip, LIDAL_PHASE_STAMPS), the prologue and the epilogue of a tile -- means
per phase in ns at the measured clock, shares of a workgroup's residency, percentiles over workgroups."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402

LAYERS = [(1, 96, 96), (8, 256, 256), (2, 32, 32), (1, 32, 32), (4, 128, 128)]


def main():
    dev = torch.device('cuda')
    L = B.lib_handle()
    L.lidal_debug_phase_stamps.restype = ctypes.c_int
    L.lidal_debug_phase_stamps.argtypes = [ctypes.c_void_p]
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    coords = torch.from_numpy(batch['coords_v_b']).to(dev)
    levels = {1: coords}
    s = 1
    while s < 16:
        levels[s * 2] = F.spdownsample(levels[s], 2, 2, s)
        s *= 2
    from lidal_amd.nn.functional.conv import _weight_image
    out_all = {}
    for stride, ci, co in LAYERS:
        c = levels[stride]
        kmap, _ = F.build_kernel_map(c, (stride,) * 3, (3, 3, 3), (1, 1, 1))
        n, m = c.shape[0], kmap.total
        o = kmap.order_out
        g = torch.Generator(device='cpu').manual_seed(ci * 1000 + co)
        x = torch.randn(n, ci, generator=g).to(dev).to(torch.bfloat16)
        w = (torch.randn(27, ci, co, generator=g) * 0.05).to(dev)
        img = _weight_image(w, torch.bfloat16, n, 0)
        y = torch.empty((n, co), dtype=torch.bfloat16, device=dev)
        tiles = (n + 127) // 128
        # column blocks of the launch: the tiling policy's (conv_img.hip pick_tiling)
        nb = 2 if co <= 32 else (4 if (co <= 64 or (co % 64 == 0 and tiles * ((co + 127) // 128) <= 384)) else
                                 (6 if (co % 128 != 0 and (co % 96 == 0 or co < 128)) else 8))
        nblk = (co + 16 * nb - 1) // (16 * nb)
        stamps = torch.zeros((nblk * tiles, 8, 12), dtype=torch.int64, device=dev)

        def launch():
            B.check(L.lidal_conv_apply_image(B.ptr(x), B.ptr(img), B.ptr(o.table), B.ptr(o.perm), B.ptr(o.tile_masks),
                                             B.ptr(y), n, n, ci, co, 27, 0, 1, None, None, 0, None, None, B.stream()), 'conv')
        assert L.lidal_debug_phase_stamps(None) == 0
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us_plain = e0.elapsed_time(e1) * 1e3 / 20
        assert L.lidal_debug_phase_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        launch()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us_stamped = e0.elapsed_time(e1) * 1e3 / 20
        assert L.lidal_debug_phase_stamps(None) == 0
        st = stamps.cpu().numpy().astype(np.float64)
        used = st[:, :, 1] > 0
        if not used.any():
            print('s%d %d->%d: no stamps (the launch did not take the lean kernel)' % (stride, ci, co))
            continue
        # clock: shader-clock ticks per 10 ns tick of s_memrealtime, from the whole-kernel pair of every wave
        ratio = st[:, :, 1][used].sum() / st[:, :, 9][used].sum()
        ns = 10.0 / ratio                           # ns per shader-clock tick
        ph = st[:, :, 0][used]
        steady = np.maximum(ph - 1, 0)              # stamped phases: all but the peeled last one
        names = ['issue (index + slab DMA + gathers, incl. wait for index(p+1))', 'wait for A(p)',
                 'fragment reads + MFMAs', 'wait for slab(p+1)', 'barrier']
        sums = [st[:, :, 3 + i][used].sum() for i in range(5)]
        tot_phase = sum(sums)
        whole = st[:, :, 1][used].sum()
        pro, epi = st[:, :, 2][used].sum(), st[:, :, 8][used].sum()
        per_wg_phase_ns = (st[:, :, 3:8].sum(2) * used).sum(1) / np.maximum((np.maximum(st[:, :, 0] - 1, 0) * used).sum(1), 1) * ns
        rec = {'rows': int(n), 'rules': int(m), 'tiles': int(tiles), 'column_blocks': int(nblk),
               'launch_us_instrumented_lib_stamps_off': round(us_plain, 2), 'launch_us_stamps_on': round(us_stamped, 2),
               'shader_clock_MHz': round(ratio * 100.0, 1),
               'phases_per_workgroup_mean': round(float(ph.mean()), 2), 'phases_per_workgroup_max': int(ph.max()),
               'ns_per_phase_mean': round(float(tot_phase / max(steady.sum(), 1) * ns), 1),
               'ns_per_phase_by_interval': {nm: round(float(v / max(steady.sum(), 1) * ns), 1) for nm, v in zip(names, sums)},
               'share_of_wave_residency': {'prologue (tile mask, index x2, first slab, first gathers, wait)': round(float(pro / whole), 4),
                                           'stamped phases': round(float(tot_phase / whole), 4),
                                           'last phase + write-out (+ BatchNorm tile statistics)': round(float(epi / whole), 4)},
               'wave_residency_us_mean': round(float(st[:, :, 1][used].mean() * ns / 1e3), 2),
               'ns_per_phase_percentiles_over_workgroups': {str(q): round(float(np.percentile(per_wg_phase_ns[per_wg_phase_ns > 0], q)), 1)
                                                             for q in (5, 25, 50, 75, 95)} if (per_wg_phase_ns > 0).any() else None}
        out_all['s%d_%d_%d' % (stride, ci, co)] = rec
        print('s%-2d %3d->%-3d rows %d rules %d  %.1f us (stamps on %.1f)  clock %.0f MHz' % (stride, ci, co, n, m, us_plain, us_stamped, ratio * 100))
        print('   phases/workgroup %.2f (max %d); %.0f ns per phase:' % (ph.mean(), ph.max(), rec['ns_per_phase_mean']))
        for nm in names:
            print('      %-70s %7.1f ns' % (nm, rec['ns_per_phase_by_interval'][nm]))
        print('   residency of a wave %.2f us: %s' % (rec['wave_residency_us_mean'], rec['share_of_wave_residency']))
        print('   ns per phase over workgroups (5/25/50/75/95 %%): %s' % rec['ns_per_phase_percentiles_over_workgroups'], flush=True)
    if len(sys.argv) > 1:
        json.dump(out_all, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
