# usage: tools/run_ab.sh OUT lib1 lib2 ...: frame_stats of two scenes per library variant
out=$1; shift
for s in dog_room dogs_and_cat; do for lib in "$@"; do echo "== $s $lib"; SCENE=$s RT_GPU_LIB=$PWD/$lib PLAIN_ONLY=1 timeout -k 10 120 python3 -u tools/frame_stats.py 2>&1 | grep -v amdgpu.ids || exit 1; done; done > gpurun_out/$out 2>&1
cat gpurun_out/$out
