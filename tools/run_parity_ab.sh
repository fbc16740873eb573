# usage: tools/run_parity_ab.sh OUT lib1 lib2 ... : quick bit-parity tests of the default library, then frame_stats of two scenes per library
out=$1; shift
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 240 --timeout-method thread -k "render_region_equals_oracle or path_length_cap or several_frames or find_ray_hit_matches" > gpurun_out/$out.tests.log 2>&1 || { tail -30 gpurun_out/$out.tests.log; exit 1; }
tail -2 gpurun_out/$out.tests.log
bash tools/run_ab.sh $out "$@"
