#!/bin/bash
# Developer A/B builds of libhare_hip: tools/build_variants.sh "name:-DFOO=1 -DBAR=2" ...  ->  hare_amd/libhare_hip_<name>.so
# (objects under hare_amd/csrc/build/v_<name>; the .so files are git-ignored but travel to the GPU box)
here=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  make -C "$here/hare_amd/csrc" -s all KDEFS="$defs" BUILD=build/v_$name OUT=../libhare_hip_$name.so 2>&1 | head -5 &
done
wait
ls -la "$here"/hare_amd/*.so
