#!/bin/bash
# Library of another commit as an A/B variant:  tools/build_ref_variant.sh <commit> <name>  ->  tools/_variants/<name>/libhessgpu.so
# (a scratch worktree under /tmp; the variant is selected at run time with HESS_LIB=...)
set -e
C=$1; N=$2
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/hess_wt_$N
rm -rf $W; git -C $R worktree prune; git -C $R worktree add -f --detach $W $C > /dev/null
(cd $W && python -m hessgpu_amd.build > /dev/null)
mkdir -p $R/tools/_variants/$N
cp $W/hessgpu_amd/libhessgpu.so $W/hessgpu_amd/libsiftgpu.so $R/tools/_variants/$N/
git -C $R worktree remove --force $W
echo "built $N from $(git -C $R rev-parse --short $C)"
