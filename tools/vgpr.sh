#!/bin/bash
# register / scratch use of the kernels of one object: tools/vgpr.sh k_step3d_t [filter] [objdir]
# (reads the notes of the gfx950 code object inside roms_trunk_mgh_amd/csrc/_obj/<name>.o)
set -e
O=${3:-roms_trunk_mgh_amd/csrc/_obj}/$1.o
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $O $T/fat.bin
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.o --unbundle
$L/llvm-readelf --notes $T/dev.o | awk '/\.name:/{n=$2} /\.private_segment_fixed_size:/{p=$2} /\.sgpr_count:/{s=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "scratch", p, "spill", $2}' | grep "${2:-.}" | c++filt | sed 's/(anonymous namespace):://g'
rm -rf $T
