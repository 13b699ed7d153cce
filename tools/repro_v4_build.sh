#!/bin/bash
# Rebuilds the round-1 event of DESIGN.md 5.2 (wrong values from the float32 FAST plain kernels
# with 4 pixels per thread): the sources of the end of round 1 (commit 94ef908) into
# build_variants/r1/, patched to instantiate the 4-pixel instances (-DMOD16_REPRO_V4), and the
# library built with several compiler settings. Runs here (hipcc cross-compiles); then, on the GPU:
#
#   R=build_variants/r1
#   MOD16_LIB=$R/libs_ref.so python $R/tools/repro_v4.py dump /tmp/ref.npz
#   for v in v4 v4_noagpr v4_nosgpr2vgpr v4_O2 v4_O1 v4_nomisched; do
#     MOD16_LIB=$R/libs_$v.so python $R/tools/repro_v4.py dump /tmp/$v.npz
#     python tools/repro_v4.py compare /tmp/ref.npz /tmp/$v.npz; done
set -e
cd "$(dirname "$0")/.."
R=build_variants/r1
rm -rf $R && mkdir -p $R
git archive 94ef908 | tar -x -C $R
rm -rf $R/profiles $R/tests $R/PAPERS.md $R/SNIPPETS.md $R/SURVEY.md
cp tools/repro_v4.py $R/tools/repro_v4.py
python3 - <<'PY'
p = 'build_variants/r1/mod16_amd/csrc/mod16_capi.hip'
s = open(p).read()
s = s.replace("    constexpr bool kFastOk = !(std::is_same<T, float>::value && V == 4);",
              "#ifdef MOD16_REPRO_V4\n    constexpr bool kFastOk = true;\n#else\n"
              "    constexpr bool kFastOk = !(std::is_same<T, float>::value && V == 4);\n#endif")
old = ("        if (fast && std::is_same<T, float>::value)\n"
       "            launch_variant<T, 2>(b, lut, fast, sep, dense, grid_for(ctx, nbody / 2), st);\n"
       "        else\n")
assert old in s
s = s.replace(old, "#ifndef MOD16_REPRO_V4\n" + old + "#endif\n")
open(p, 'w').write(s)
PY
F="--offload-arch=gfx950 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-unused-function"
SRC=$R/mod16_amd/csrc/mod16_capi.hip
build() { /opt/rocm/bin/hipcc $F "${@:2}" -o $R/libs_$1.so $SRC; }
build ref -O3 &
build v4 -O3 -DMOD16_REPRO_V4 &
build v4_noagpr -O3 -DMOD16_REPRO_V4 -mllvm -amdgpu-spill-vgpr-to-agpr=0 &
build v4_nosgpr2vgpr -O3 -DMOD16_REPRO_V4 -mllvm -amdgpu-spill-sgpr-to-vgpr=0 &
wait
build v4_O2 -O2 -DMOD16_REPRO_V4 &
build v4_O1 -O1 -DMOD16_REPRO_V4 &
build v4_nomisched -O3 -DMOD16_REPRO_V4 -mllvm -enable-misched=0 &
wait
ls -la $R/*.so
