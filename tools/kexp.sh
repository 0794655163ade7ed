#!/bin/bash
# One in-step kernel-time measurement under rocprofv3 (GPU box, repo root):  bash tools/kexp.sh <label> <kernel substrings...> -- VAR=val ... <script.py>
# prints "<label> <kernel> calls avg min max" for every kernel whose name contains one of the substrings
LABEL=$1; shift
SUBS=()
while [ "$1" != "--" ]; do SUBS+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/kexp_$$_$RANDOM
( cd /tmp && export TMPDIR=/tmp && env "${@:1:$#-1}" rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $REPO/${@: -1} > $D.log 2>&1 ) || tail -5 $D.log
python $REPO/tools/kstat.py $D "${SUBS[@]}" | sed "s/^/$LABEL /"
rm -rf $D $D.log
