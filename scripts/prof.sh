#!/bin/bash
# prof.sh TAG [VAR=value ...] -- command ... : run `command` under rocprofv3 --kernel-trace --stats (per-kernel durations) with the
# given environment, leaving gpurun_out/prof/TAG_kernel_stats.csv and printing its top rows.  The caller copies what it wants
# judged into profiles/.
tag=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_$tag -o $tag -- "$@" > $out/$tag.stdout 2> $out/$tag.stderr
f=$(find $out/raw_$tag -name "${tag}_kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/${tag}_kernel_stats.csv; echo "== $tag ${envs[*]}"; head -12 $out/${tag}_kernel_stats.csv | cut -c1-220; fi
rm -rf $out/raw_$tag
