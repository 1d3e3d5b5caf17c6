"""GPU: wall time of the 2-rank scoring child processes, all-gather vs halo exchange."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import test_multirank_gpu as t  # noqa: E402

for mode in ('score_allgather', 'score', 'score'):
    d = tempfile.mkdtemp()
    t0 = time.time()
    t._spawn(mode, d)
    print(mode, '%.1f s' % (time.time() - t0), flush=True)
