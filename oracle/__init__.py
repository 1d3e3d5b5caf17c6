"""oracle/ -- CPU restatement of the LiDAL sparse-voxel hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (lidal_amd/) may import from here.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and there only as the
checker / the timed CPU baseline, never as the product path.

What it restates
  * oracle.tsref      -- the torchsparse==1.4.0 operator API exactly as the reference uses it
                         (/root/reference/docs/requirements.txt:191, network/utils.py:13-102,
                         network/spvcnn.py:112-155, network/minkunet.py:97-122).  torchsparse is a
                         third-party dependency that is NOT vendored in /root/reference and not
                         installed here, so this follows its published v1.4.0 algorithm.
                         **parity unpinned** against torchsparse itself (no reference test or golden
                         vector exists for it); pinned instead by spec-derived
                         hash known-answer values, algebraic properties (dense-grid conv equals
                         torch.nn.functional.conv3d, ...), gradcheck, and by running the reference's
                         own network/*.py unchanged on top of it (tests/golden/make_golden.py).
  * oracle.scoring_ref -- score/sv_level/LiDAL.py:27-103 (worker_func) and :225-330 (selection),
                         restated on in-memory arrays.  PINNED: checked against the reference's
                         own worker_func / __main__ run in the build container
                         (tests/golden/make_golden.py -> tests/golden/scoring_*.npz).
  * oracle.harness_ref -- train step (train.py:121-156) and inference post-processing
                         (score/prob_inference.py:97-113) on the CPU restatement.
"""
