# the model-level GPU tests once more with EVERY Model on two streams under the 'tail' policy (pairing / grouping / R13 / R14 under
# other plans than the speed protocol's): failures that are not about the stream count itself would be real
mkdir -p gpurun_out
DEEPHAR_STREAMS=2 DEEPHAR_STREAM_POLICY=tail python -m pytest tests/test_gpu_models.py tests/test_gpu_spnet_flat.py tests/test_gpu_real_configs.py tests/test_gpu_full_configs.py tests/test_gpu_caller_loop.py tests/test_gpu_action_loops.py tests/test_gpu_speed2d.py -q -m gpu 2>&1 | tail -25 > gpurun_out/two_stream_suite.log; tail -25 gpurun_out/two_stream_suite.log
