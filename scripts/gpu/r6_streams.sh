#!/bin/bash
# round 6: the streamed weight gradient against the offset-major one (scripts/exp/wgrad_streams.py)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_streams; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python scripts/exp/wgrad_streams.py 2>&1 | grep -v amdgpu.ids | tail -7; }
{
run LEVEL=0 KEY=parent BLOCK=1024 W=512 LIDAL_X_STREAM_DEPTH=1
run LEVEL=0 KEY=parent BLOCK=1024 W=512
run LEVEL=1 KEY=row BLOCK=1024 W=512 LIDAL_X_STREAM_DEPTH=1
run LEVEL=2 KEY=row BLOCK=1024 W=512 LIDAL_X_STREAM_DEPTH=1 CA=128 CB=128
run LEVEL=2 KEY=row BLOCK=1024 W=512 CA=64 CB=128
} > $O/log.txt 2>&1
cat $O/log.txt
