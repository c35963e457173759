#!/bin/bash
# Manual tool: the gz layer's numbers for profiles/: gzwrite()/gzread() of a large file in pieces of several sizes, with
# and without the chunk index.  usage: bash tests/gpu_gz_profile.sh <commit> > profiles/rNN_gz_layer.txt
set -e
echo "# tests/gpu_gz_bench.py at commit ${1:-?}: mixed corpus, level 6, file in the page cache of the box (tmpfs or disk cache);"
echo "# every line is one whole pass (gzopen .. gzclose), two passes each.  'I' in the mode: the chunk index and its split records."
for cfg in "3584 512 wb6I" "2048 1024 wb6I" "2048 64 wb6I" "2048 16 wb6I" "2048 1 wb6I" "2048 512 wb6" "2048 64 wb6"; do
    echo "## $cfg"
    timeout -k 10 200 python3 -u tests/gpu_gz_bench.py $cfg
done
echo "## 2048 64 wb6I with ZAMD_GZ_BEHIND=0 (writes compressed on the caller's thread)"
ZAMD_GZ_BEHIND=0 timeout -k 10 200 python3 -u tests/gpu_gz_bench.py 2048 64 wb6I
