set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/bench_ops.py > gpurun_out/r02/ops_GBs.jsonl 2> gpurun_out/r02/ops.err
python tools/bench_bwd.py 4 64 > gpurun_out/r02/backward_ms.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_bwd.py 4 128 >> gpurun_out/r02/backward_ms.jsonl 2>> gpurun_out/r02/ops.err
python g-nerf_amd/train_step_mi355x.py --steps 10 > gpurun_out/r02/train_step.jsonl 2>> gpurun_out/r02/ops.err
python g-nerf_amd/train_step_mi355x.py --steps 10 --force-fp32 >> gpurun_out/r02/train_step.jsonl 2>> gpurun_out/r02/ops.err
python g-nerf_amd/train_step_mi355x.py --mode renderer --steps 20 >> gpurun_out/r02/train_step.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_generator.py > gpurun_out/r02/generator.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_shapes.py > gpurun_out/r02/shapes.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_orbit.py > gpurun_out/r02/orbit.jsonl 2>> gpurun_out/r02/ops.err
tail -n 3 gpurun_out/r02/*.jsonl
tail -5 gpurun_out/r02/ops.err
