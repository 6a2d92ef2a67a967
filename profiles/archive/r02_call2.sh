# round 2, call 2: joint labeled+unlabeled pass (two BatchNorm segments per launch) - device tests of the new code, then A/B vs the two-pass step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_segmented_bn.py tests/test_emu_engine.py tests/test_emu_conv.py tests/test_emu_tracker.py tests/test_configs_c1_c5.py tests/test_gpu_fullsize.py -q -m gpu --timeout 240 -p no:cacheprovider 2>&1 | tail -40) > gpurun_out/r02b_pytest_gpu.log; tail -3 gpurun_out/r02b_pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r02b_bench_joint.json.log 2>&1; tail -1 gpurun_out/r02b_bench_joint.json.log | cut -c1-330
LP_JOINT_FORWARD=0 timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02b_bench_twopass.json.log 2>&1; tail -1 gpurun_out/r02b_bench_twopass.json.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 > gpurun_out/r02b_bench_joint_noprofile.json.log 2>&1; tail -1 gpurun_out/r02b_bench_joint_noprofile.json.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --size 256 > gpurun_out/r02b_bench_joint_256.json.log 2>&1; tail -1 gpurun_out/r02b_bench_joint_256.json.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --no-profile --steps 10 --backbone vits_dino > gpurun_out/r02b_bench_joint_vit.json.log 2>&1; tail -1 gpurun_out/r02b_bench_joint_vit.json.log | cut -c1-330
LP_DUMP_LAUNCHES=gpurun_out/r02b_launches.json timeout 300 python bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02b_prof -o joint -- python bench.py --no-cpu-baseline --no-profile --steps 8 --warmup 2 > gpurun_out/r02b_prof.log 2>&1; ls gpurun_out/r02b_prof | head
LP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02b_prof_serial -o serial -- python bench.py --no-cpu-baseline --no-profile --steps 8 --warmup 2 > gpurun_out/r02b_prof_serial.log 2>&1
