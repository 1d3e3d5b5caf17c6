"""GPU: time of lidal_kmap_order (row mask -> sort -> permuted table -> tile masks) per level of the
bench batch, 3x3x3 maps (27-bit masks: the merge-sort path).  A/B libraries via LIDAL_AMD_LIB."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from lidal_amd import backend as B, synth  # noqa: E402
from lidal_amd.nn import functional as F  # noqa: E402
from lidal_amd.nn.functional import conv as C  # noqa: E402
from exp_img import timeit  # noqa: E402


def main():
    print('lib', B.LIB_PATH)
    batch = synth.make_train_batch(n_frames=5, n_points=120000, seed=7122)
    c = torch.from_numpy(batch['coords_v_b']).cuda()
    s = 1
    while s <= 16:
        kmap, _ = F.build_kernel_map(c, (s,) * 3, (3, 3, 3), (1, 1, 1))
        t = timeit(lambda: C.RowOrder(kmap.nbr_out))
        print('stride %2d  %7d rows  kmap_order %7.1f us' % (s, c.shape[0], t), flush=True)
        if s < 16:
            c = F.spdownsample(c, 2, 2, s)
        s *= 2


if __name__ == '__main__':
    main()
