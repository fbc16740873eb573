# the benchmark frame of two scenes with the reference's tree and with the surface-area builder's at several leaf sizes
set -e
export PLAIN_ONLY=1
for scene in dog_room dogs_and_cat; do
echo "== $scene reference tree"
SCENE=$scene timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids
for leaf in 16 8 4 2; do
echo "== $scene sah leaf $leaf"
SCENE=$scene TREE=sah LEAF=$leaf timeout -k 10 120 python3 tools/frame_stats.py 2>&1 | grep -v amdgpu.ids
done; done
